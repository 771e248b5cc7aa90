// Host-side check of the bank read's launch plan (rmnet_amd/csrc/common.h: bank_chunks; rmnet_amd/csrc/bank.hip: the
// chunk -> segment walk of bk_main and pair_slots).  No GPU, no HIP API call: hipcc only supplies the __host__ __device__
// qualifiers of common.h.  tests/test_host_logic.py builds and runs it.
//
// The chunk walk and pair_slots() are device code inside bank.hip's anonymous namespace and cannot be included, so they
// are RESTATED here, statement by statement (bank.hip "aligned chunk" / "remainder chunk" and pair_slots); what is under
// test is bank_chunks() itself -- the real function -- and the invariants the kernel relies on:
//   1. every (query tile, memory tile) of an object is walked by exactly one segment;
//   2. every segment's partial slot is unique and below the object's slot budget nch + (R > 0 ? nqt : 0);
//   3. the segments of a pair sit at positions 0 .. count-1 of the pair's slot list, and that list (the closed form the
//      last arriver merges from) names exactly their slots;
//   4. a chunk never starts more segments than the merge scratch allows, no chunk is longer than C tile units;
//   5. the plan's search for C ends with no more chunks than workgroups, with and without "own column blocks", and with the
//      EQUALISED blocks (no remainder chunks; block lengths differ by at most one tile and never exceed C);
//   6. [r6] a launch with more pairs than workgroups runs in ROUNDS: whole objects first (one column block each), the objects behind them cut
//      into equal blocks that fill one more round; no remainder chunks, slots only for the cut objects, at most R + 2 rounds;
//   7. plan_div's DEVICE branch (an fp32 reciprocal, off by at most one, fixed up by the remainder) restated with the reciprocal one ulp
//      low / correctly rounded / one ulp high (v_rcp_f32 is accurate to one ulp): exact for every operand pair below its documented 2^22
//      (round-5 advisor: only the host branch x / c was under test).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>

#include "../../rmnet_amd/csrc/common.h"

using namespace rmnet;

namespace {

struct Seg { int qt, j0, n, slot, sself; };

// bank.hip, compute(): the segments of chunk `cl` of one object (slot_obj = 0)
std::vector<Seg> chunk_segments(const BankChunks& bc, int nqt, int njt, int cl) {
  std::vector<Seg> out;
  const int C = bc.C;
  if (cl < nqt * bc.nfull) {
    const int blk = cl / nqt;
    int j0, n;
    bank_block_range(bc, njt, blk, j0, n);
    out.push_back({cl - blk * nqt, j0, n, cl, blk});
    return out;
  }
  const int cr = cl - nqt * bc.nfull;
  const int u0 = cr * C, u1 = u0 + C, span = bc.R + bc.sc;
  for (int qt = u0 / span; qt < nqt && qt * span < u1; ++qt) {
    const int j0 = std::max(u0 - qt * span, 0), j1 = std::min(u1 - qt * span, bc.R);
    if (j1 <= j0) continue;
    out.push_back({qt, bc.nfull * bc.Cb + j0, j1 - j0, nqt * bc.nfull + cr + qt, bc.nfull + cr - (qt * span) / C});
  }
  return out;
}

// bank.hip, pair_slots(): slot list of pair qt (sb = 0)
std::vector<int> pair_slot_list(const BankChunks& bc, int nqt, int qt) {
  std::vector<int> s;
  for (int i = 0; i < bc.nfull; ++i) s.push_back(qt + i * nqt);
  if (bc.R > 0) {
    const int v0 = qt * (bc.R + bc.sc);
    const int cf = v0 / bc.C, cl = (v0 + bc.R - 1) / bc.C;
    for (int c = cf; c <= cl; ++c) s.push_back(nqt * bc.nfull + c + qt);
  }
  return s;
}

int fails = 0;
#define CHECK(cond, ...)                                   \
  do {                                                     \
    if (!(cond)) {                                         \
      if (fails < 20) { std::printf("FAIL %s: ", #cond); std::printf(__VA_ARGS__); std::printf("\n"); } \
      ++fails;                                             \
    }                                                      \
  } while (0)

void check_object(int nqt, int njt, int C, int sc, int own) {
  const BankChunks bc = bank_chunks(nqt, njt, C, sc, own);
  if (nqt == 0 || njt == 0) {
    CHECK(bc.nch == 0 || njt == 0 || nqt == 0, "nqt %d njt %d C %d", nqt, njt, C);
    CHECK(bc.nch >= 0 && bc.nrem >= 0, "negative chunk count: nqt %d njt %d C %d sc %d -> nch %d nrem %d", nqt, njt, C, sc, bc.nch, bc.nrem);
    if (nqt == 0) CHECK(bc.nch == 0, "an object without query tiles has chunks: njt %d C %d sc %d nch %d", njt, C, sc, bc.nch);
    return;
  }
  CHECK(bc.nch == nqt * bc.nfull + bc.nrem, "nch");
  if (!bc.eq) CHECK(bc.nfull * bc.Cb + bc.R == njt, "tiles: nfull %d Cb %d R %d njt %d", bc.nfull, bc.Cb, bc.R, njt);
  if (bc.eq) {
    CHECK(bc.R == 0 && bc.nrem == 0 && bc.Cb <= C, "equalised plan with a remainder / a block longer than C: njt %d C %d Cb %d", njt, C, bc.Cb);
    for (int b = 0; b < bc.nfull; ++b) {
      int j0, n;
      bank_block_range(bc, njt, b, j0, n);
      CHECK(n >= 1 && n <= bc.Cb && n >= bc.Cb - 1, "equalised block %d of %d: %d tiles (longest %d, njt %d)", b, bc.nfull, n, bc.Cb, njt);
    }
  }
  const int nsl = bc.nch + (bc.R > 0 ? nqt : 0);
  std::vector<std::vector<int>> cover(nqt, std::vector<int>(njt, 0));
  std::set<int> slots;
  std::map<int, std::vector<Seg>> by_pair;
  for (int cl = 0; cl < bc.nch; ++cl) {
    const std::vector<Seg> segs = chunk_segments(bc, nqt, njt, cl);
    int units = 0;
    for (const Seg& s : segs) {
      CHECK(s.n > 0 && s.j0 >= 0 && s.j0 + s.n <= njt, "segment range qt %d j0 %d n %d (njt %d C %d)", s.qt, s.j0, s.n, njt, C);
      for (int j = s.j0; j < s.j0 + s.n && j < njt; ++j) ++cover[s.qt][j];
      CHECK(s.slot >= 0 && s.slot < nsl, "slot %d outside [0, %d): nqt %d njt %d C %d sc %d own %d", s.slot, nsl, nqt, njt, C, sc, (int)own);
      CHECK(slots.insert(s.slot).second, "slot %d used twice: nqt %d njt %d C %d sc %d", s.slot, nqt, njt, C, sc);
      by_pair[s.qt].push_back(s);
      units += s.n;
    }
    CHECK(units <= C, "chunk %d walks %d tiles > C %d", cl, units, C);
  }
  for (int qt = 0; qt < nqt; ++qt)
    for (int j = 0; j < njt; ++j)
      CHECK(cover[qt][j] == 1, "pair %d tile %d covered %d times: nqt %d njt %d C %d sc %d own %d", qt, j, cover[qt][j], nqt, njt, C, sc, (int)own);
  for (int qt = 0; qt < nqt; ++qt) {
    const std::vector<int> list = pair_slot_list(bc, nqt, qt);
    std::vector<Seg>& segs = by_pair[qt];
    CHECK(segs.size() == list.size(), "pair %d: %zu segments, slot list of %zu: nqt %d njt %d C %d sc %d", qt, segs.size(), list.size(), nqt, njt, C, sc);
    CHECK((int)list.size() <= kSplitMax, "pair %d has %zu partials > kSplitMax", qt, list.size());
    std::set<int> pos;
    for (const Seg& s : segs) {
      CHECK(s.sself >= 0 && s.sself < (int)list.size(), "sself %d outside the pair's list of %zu", s.sself, list.size());
      if (s.sself >= 0 && s.sself < (int)list.size())
        CHECK(list[s.sself] == s.slot, "pair %d position %d: list says slot %d, the segment owns %d (nqt %d njt %d C %d sc %d own %d)", qt, s.sself,
              list[s.sself], s.slot, nqt, njt, C, sc, (int)own);
      CHECK(pos.insert(s.sself).second, "position %d of pair %d taken twice", s.sself, qt);
    }
  }
}

// bank.hip, the plan
void check_launch(const std::vector<int>& nqt, const std::vector<int>& njt, int target, int sc, int cq) {
  long long W = 0;
  int njt_max = 0;
  for (size_t o = 0; o < nqt.size(); ++o) { W += (long long)nqt[o] * njt[o]; njt_max = std::max(njt_max, njt[o]); }
  auto total = [&](int C, int own) { int n = 0; for (size_t o = 0; o < nqt.size(); ++o) n += bank_chunks(nqt[o], njt[o], C, sc, own).nch; return n; };
  // [r6] more pairs than workgroups: ROUNDS of aligned chunks (common.h: bank_round_chunk_len; bank.hip, the plan)
  {
    int P = 0, nq_max = 0;
    for (size_t o = 0; o < nqt.size(); ++o) { P += njt[o] > 0 ? nqt[o] : 0; nq_max = std::max(nq_max, nqt[o]); }
    if (P > target) {
      const int R = P / target;
      int Pw = 0, ps = 0;
      for (size_t o = 0; o < nqt.size(); ++o) { ps += njt[o] > 0 ? nqt[o] : 0; if (ps <= R * target) Pw = std::max(Pw, ps); }
      long long N = 0, slots = 0;
      bool split_seen = false;
      ps = 0;
      for (size_t o = 0; o < nqt.size(); ++o) {
        ps += njt[o] > 0 ? nqt[o] : 0;
        const int Co = bank_round_chunk_len(njt[o], ps, P, Pw, target, cq);
        const BankChunks bc = bank_chunks(nqt[o], njt[o], Co, sc, 2);
        CHECK(bc.R == 0 && bc.nrem == 0, "rounds: object %zu has remainder chunks", o);
        if (njt[o] > 0 && nqt[o] > 0) {
          if (ps <= R * target) {
            CHECK(bc.nfull == 1 && !split_seen, "rounds: object %zu inside the whole rounds is cut in %d blocks (or follows a split object)", o, bc.nfull);
          } else {
            split_seen = true;
          }
        }
        N += bc.nch;
        slots += bc.nfull > 1 ? bc.nch : 0;
        check_object(nqt[o], njt[o], Co, sc, 2);
      }
      CHECK(N <= (long long)(R + 2) * target + nq_max, "rounds: %lld chunks on %d workgroups are more than %d rounds (P %d)", N, target, R + 2, P);
      CHECK(slots <= (long long)kSplitTargetSlots + (long long)nqt.size() * nq_max, "rounds: %lld partial slots exceed the workspace of a launch group", slots);
      return;
    }
  }
  // [r6] pairs fit the workgroups: bank_plan_pick() over the equalised plan and the plain plan, priced as bank.hip's
  // planning wave prices them (lane i = candidate i)
  const int cmin = (bank_chunk_min(njt_max) + cq - 1) / cq * cq;
  int Clo = std::max((int)((W + target - 1) / target), cmin);
  Clo = (Clo + cq - 1) / cq * cq;
  const int Chi = std::max((njt_max + cq - 1) / cq * cq, Clo);
  const int step = (std::max((Chi - Clo + 62) / 63, 1) + cq - 1) / cq * cq;
  int c1 = 0, ce[64], cp[64], lp = -1;
  for (int i = 0; i < 64; ++i) {
    ce[i] = bank_eq_candidate(i, njt_max, Clo, cq, cmin);
    cp[i] = Clo + i * step;
    int me = 0;
    for (size_t o = 0; o < nqt.size(); ++o) me += bank_eq_count(nqt[o], njt[o], ce[i]);
    CHECK(me == total(ce[i], 2), "bank_eq_count and bank_chunks(.., 2) disagree at c %d: %d vs %d", ce[i], me, total(ce[i], 2));
    if (me <= target && (c1 == 0 || ce[i] < c1)) c1 = ce[i];
    if (lp < 0 && total(cp[i], 0) <= target) lp = i;
  }
  const int cw = ce[0];
  CHECK(total(cw, 2) <= target, "whole pairs (%d chunks) do not fit %d workgroups although P <= target", total(cw, 2), target);
  CHECK(c1 > 0, "no one-round equalised plan although the pairs fit the workgroups (target %d)", target);
  BankPlanPick pk = bank_plan_pick(c1, cw, Clo, sc, sc == kSegCost ? 3 : 1);   // (against the even cut first: bank.hip)
  if (pk.blocks == 0) pk = bank_plan_pick(c1, cw, lp >= 0 ? cp[lp] : 0, sc, sc == kSegCost ? 3 : 1);
  const int C0 = pk.C > 0 ? pk.C : Chi;
  int blocks = pk.blocks;
  if (!blocks && total(C0, 1) <= target) blocks = 1;
  CHECK(total(C0, blocks) <= target, "plan %d at C %d: more chunks (%d) than workgroups (%d)", blocks, C0, total(C0, blocks), target);
  {   // partial slots of the launch fit the workspace of a launch group (common.h: bank_group_slot0)
    long long slots = 0;
    int nq_max = 0;
    for (size_t o = 0; o < nqt.size(); ++o) {
      const BankChunks bc = bank_chunks(nqt[o], njt[o], C0, sc, blocks);
      slots += bc.nch + (bc.R > 0 ? nqt[o] : 0);
      nq_max = std::max(nq_max, nqt[o]);
    }
    CHECK(slots <= (long long)target + (long long)nqt.size() * nq_max, "%lld partial slots exceed the workspace of a launch group", slots);
  }
  if (blocks == 2)   // an equalised plan never cuts a pair into more partials than the merge holds
    for (size_t o = 0; o < nqt.size(); ++o) CHECK(bank_chunks(nqt[o], njt[o], C0, sc, 2).nfull <= kSplitMax, "object %zu cut %d-fold", o, bank_chunks(nqt[o], njt[o], C0, sc, 2).nfull);
  for (size_t o = 0; o < nqt.size(); ++o) check_object(nqt[o], njt[o], C0, sc, blocks);
}

}  // namespace


// common.h, plan_div(), the __HIP_DEVICE_COMPILE__ branch, with v_rcp_f32 modelled as the correctly rounded reciprocal moved by `ulp` ulps
int plan_div_device(int x, int c, int ulp) {
  float rc = 1.0f / (float)c;
  for (int i = 0; i < ulp; ++i) rc = std::nextafterf(rc, 2.0f);
  for (int i = 0; i > ulp; --i) rc = std::nextafterf(rc, 0.0f);
  int q = (int)((float)x * rc);
  const int r = x - q * c;
  return q + (r >= c ? 1 : 0) - (r < 0 ? 1 : 0);
}

void check_plan_div_device() {
  auto one = [&](int x, int c) {
    for (int ulp = -1; ulp <= 1; ++ulp)
      if (plan_div_device(x, c, ulp) != x / c) {
        if (fails < 20) std::printf("plan_div device branch: %d / %d with the reciprocal %+d ulp -> %d, exact %d\n", x, c, ulp, plan_div_device(x, c, ulp), x / c);
        ++fails;
      }
  };
  const int top = (1 << 22) - 1;
  for (int c = 1; c <= 2048; ++c)                       // every divisor the plan can see (chunk lengths, query tiles, workgroups) ...
    for (int k = 0; k <= 64; ++k) {                     // ... against numerators around its multiples, small and at the contract's edge
      for (int d = -2; d <= 2; ++d) {
        const long long lo = (long long)k * c + d, hi = (long long)(top / c - k) * c + d;
        if (lo >= 0 && lo <= top) one((int)lo, c);
        if (hi >= 0 && hi <= top) one((int)hi, c);
      }
    }
  std::mt19937 rng(99);
  for (int i = 0; i < 4000000; ++i) {
    const int c = 1 + (int)(rng() % (i & 1 ? top : 4096)), x = (int)(rng() % (unsigned)(top + 1));
    one(x, c);
  }
}

int main() {
  // 1. bank_chunks on a dense sweep of single objects (both segment costs, with and without own blocks)
  for (int sc : {kSegCost, kSegCostF16})
    for (int nqt = 0; nqt <= 14; ++nqt)
      for (int njt = 0; njt <= 150; njt += (njt < 40 ? 1 : 7))
        for (int C = 4; C <= 160; C += (C < 30 ? 1 : 5))
          for (int own = 0; own < 3; ++own) check_object(nqt, njt, C, sc, own);
  // 2. whole launches: the shapes of the tests / the bench, and random mixes (empty objects, many equal objects)
  const int targets[] = {256, 232, 209, 192, 64};
  std::mt19937 rng(7);
  std::vector<std::pair<std::vector<int>, std::vector<int>>> launches = {
      {std::vector<int>(8, 12), std::vector<int>(8, 120)},  {std::vector<int>(16, 12), std::vector<int>(16, 120)},
      {std::vector<int>(20, 12), std::vector<int>(20, 120)}, {std::vector<int>(32, 12), std::vector<int>(32, 120)},
      {std::vector<int>(50, 4), std::vector<int>(50, 16)},   {std::vector<int>(64, 1), std::vector<int>(64, 1)},
      {{14, 13, 12, 11, 9}, {140, 120, 100, 90, 75}},        {{57, 57, 57}, {1040, 1040, 1040}}, {{26}, {120}}};
  for (int r = 0; r < 400; ++r) {
    const int ng = 1 + rng() % 64;
    std::vector<int> q(ng), j(ng);
    const int qmax = 1 + rng() % 57, jmax = 1 + rng() % (r % 3 ? 200 : 1100);
    for (int o = 0; o < ng; ++o) { q[o] = rng() % 9 == 0 ? 0 : 1 + rng() % qmax; j[o] = rng() % 11 == 0 ? 0 : 1 + rng() % jmax; }
    launches.push_back({q, j});
  }
  for (const auto& l : launches)
    for (int t : targets) {
      check_launch(l.first, l.second, t, kSegCost, 1);
      check_launch(l.first, l.second, t, kSegCostF16, 2);
    }
  check_plan_div_device();
  if (fails) { std::printf("plan_check: %d failures\n", fails); return 1; }
  std::printf("plan_check ok: %zu launches x %zu targets x 2 modes\n", launches.size(), sizeof(targets) / sizeof(targets[0]));
  return 0;
}
