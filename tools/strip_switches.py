# -*- coding: utf-8 -*-
"""Resolve the experiment switches of csrc/bank.hip (BK_ABLATE / BK_TRACE / BK_TAIL / ... at their default values) and print
the switch-free source: the product kernel without ~45 `#if BK_*` sites.  The inverse diff is kept as
tools/patches/bank_experiment_switches.patch (`git apply` it to get the switches back for an ablation / trace build;
tools/build_variant.sh does so on a scratch copy with PATCH=...).  The product's ISA must not change: compare
`hipcc -S --cuda-device-only` of both (tools/README.md).     python tools/strip_switches.py in.hip > out.hip"""
import re, sys

DEFAULTS = {'BK_TRACE': 0, 'BK_ABLATE': 0, 'BK_TAIL': 0, 'BK_STATIC_ABL': 0, 'BK_PRIO': 2, 'BK_SCHED': 1, 'BK_YPRIO': 0, 'BK_VNT': 0,
            'BK_F16_INTERLEAVE': 6, 'BK_PF': 7, 'BK_PF_REM': 0, 'BK_OUT_AUX': 0, 'BK_PLAN_NOEQ': 0, 'BK_CLK': 0,
            'BK_STATIC_ROWS': 2, 'BK_STATIC_BPUS': '35.0e3f', 'BK_STATIC_BPUS_F16': '50.0e3f'}
KEEP_AS_CONSTANT = {   # real tuning parameters: they stay, as named constants
    'BK_PRIO': ('kProducerPrio', 'static priority of the producer waves (s_setprio)'),
    'BK_F16_INTERLEAVE': ('kSoftmaxInterleave', 'fp16 modes, producers: soft-max VALU instructions scheduled between two S MFMAs'),
    'BK_PF': ('kPfSteps', 'fp16 modes: L2 prefetch distance in steps of two tiles (L2Prefetch)'),
}
LITERALS = {'BK_STATIC_ROWS': '2', 'BK_STATIC_BPUS': '35.0e3f', 'BK_STATIC_BPUS_F16': '50.0e3f'}   # (already named constants in the source)


def cond_value(expr):
    e = expr
    for k, v in DEFAULTS.items():
        e = re.sub(r'\b%s\b' % k, str(v), e)
    if re.search(r'\bBK_|[A-Za-z_]{2,}', e.replace('defined', '')):
        return None                                   # not (only) an experiment switch: leave the directive alone
    e = e.replace('&&', ' and ').replace('||', ' or ').replace('!', ' not ').replace(' not =', '!=')
    return bool(eval(e))


def strip(lines):
    out, stack = [], []       # stack entries: [kind, emitting_before, taken, active] ; kind 'x' = untouched directive
    emitting = True
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        m = re.match(r'#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)', s)
        if not m:
            if emitting:
                out.append(ln)
            i += 1
            continue
        kw, rest = m.group(1), m.group(2).split('//')[0].strip()
        if kw in ('if', 'ifdef', 'ifndef'):
            if kw == 'if':
                v = cond_value(rest)
            else:
                v = None
                if rest in DEFAULTS or rest == 'BK_TRACE_WAVE':      # "#ifndef BK_X / #define BK_X v / #endif": drop the block
                    v = False if kw == 'ifndef' else True
                    if kw == 'ifndef' and rest == 'BK_TRACE_WAVE':
                        v = False
            if v is None:
                stack.append(['x', emitting, False, emitting])
                if emitting:
                    out.append(ln)
            else:
                stack.append(['r', emitting, v, emitting and v])
                emitting = emitting and v
        elif kw == 'else':
            top = stack[-1]
            if top[0] == 'x':
                if emitting:
                    out.append(ln)
            else:
                emitting = top[1] and not top[2]
        elif kw == 'elif':
            top = stack[-1]
            assert top[0] == 'x', 'elif on a resolved switch: not handled'
            if emitting:
                out.append(ln)
        else:   # endif
            top = stack.pop()
            if top[0] == 'x':
                if emitting:
                    out.append(ln)
            emitting = top[1]
            # comment continuation lines that belonged to a dropped "#define BK_X" block (start with // far to the right)
            if top[0] == 'r':
                while i + 1 < len(lines) and re.match(r'\s{20,}//', lines[i + 1]):
                    i += 1
        i += 1
    return out


src = open(sys.argv[1]).read().split('\n')
text = '\n'.join(strip(src))
# in-code uses of the switches at their default values
subs = [
    (r'\n[ \t]*BK_STAMP\(\);[^\n]*', ''),                                   # trace stamps (whole statement lines)
    (r' && !\(\(BK_ABLATE & 2048\) && \(n & 1\)\)', ''),
    (r'\n *if \(BK_ABLATE & 4096\) \{[^\n]*\}[^\n]*', ''),
    (r'\n *if \(BK_ABLATE & 1024\) return;', ''),
    (r' && !\(BK_ABLATE & 8\)', ''),
    (r'ceq && !\(BK_PLAN_NOEQ\)', 'ceq'),
    (r'producer && BK_PRIO > 0', 'producer'),
    (r'wk\.pf_nparts = BK_PF_REM \? 1 : 0;', 'wk.pf_nparts = 0;          // (remainder chunks do not prefetch: alone they would touch every line of a step)'),
    (r', BK_OUT_AUX\)', ', 0)'),
    (r'if \(BK_PF\) ', ''),
]
for a, b in subs:
    text = re.sub(a, b, text)
# the trace plumbing (BK_TRACE builds only)
text = re.sub(r', long long t_entry,', ',', text)
text = re.sub(r', t_entry, ', ', ', text)
text = re.sub(r'\n  const long long t_entry = [^\n]*', '', text)
text = re.sub(r'\n  int trace_slot;[^\n]*', '', text)
text = re.sub(r'\n  a\.trace_slot = [^\n]*', '', text)
for k, (name, _) in KEEP_AS_CONSTANT.items():
    text = re.sub(r'\b%s\b' % k, name, text)
for k, v in LITERALS.items():
    text = re.sub(r'\b%s\b' % k, v, text)


# `if (!(BK_ABLATE & n)) {` -> a bare scope (the blocks declare locals)
text = re.sub(r'if \(!\(BK_ABLATE & \d+\)\) \{', '{', text)
consts = ''.join('constexpr int %s = %s;%s\n' % (name, DEFAULTS[k], '   // ' + note if note else '') for k, (name, note) in KEEP_AS_CONSTANT.items())
text = text.replace('constexpr int kRThreads = 64 * (kProducers + kConsumers);  // 768 = 12 waves = 3 per SIMD\n',
                    'constexpr int kRThreads = 64 * (kProducers + kConsumers);  // 768 = 12 waves = 3 per SIMD\n' + consts, 1)
text = text.replace('#define BK_STAMP() do {} while (0)\n', '')
left = sorted(set(re.findall(r'\bBK_[A-Z0-9_]+', text)))
sys.stderr.write('switch names left in the text (comments or hand work): %s\n' % left)
sys.stdout.write(text)
