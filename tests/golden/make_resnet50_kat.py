# -*- coding: utf-8 -*-
"""Known-answer file for the ResNet-50 trunk that RMNet takes from torchvision
(models/rmnet.py:57-64, 86-94: ``conv1, bn1, layer1, layer2, layer3`` of
``torchvision.models.resnet50``).  torchvision (requirements.txt: >= 0.3.0) is not installed in this
image and its weights cannot be downloaded, so the architecture is stated here from its published
definition -- He et al. 2015, Table 1, 50-layer column, in torchvision's "v1.5" form (stride on the
3x3 convolution of the first block of a stage, torchvision/models/resnet.py ``Bottleneck``) --
independently of rmnet_amd/networks.py, which the test then compares against:

    stem   conv1 7x7/2 3->64 (no bias), bn1, relu, maxpool 3x3/2
    layer1 3 bottlenecks, width 64,  out 256,  stride 1
    layer2 4 bottlenecks, width 128, out 512,  stride 2
    layer3 6 bottlenecks, width 256, out 1024, stride 2
    bottleneck: conv1 1x1 in->width, conv2 3x3 width->width (stride s), conv3 1x1 width->4*width, each
    followed by a BatchNorm2d; the first block of a stage has downsample = (1x1 conv stride s, BN).

Known totals of the published model (used as a cross-check of this very script): conv1+bn1 9,536
parameters, layer1 215,808, layer2 1,219,584, layer3 7,098,368 (with layer4 14,964,736 and fc 2,049,000
that is torchvision's documented 25,557,032).

    python tests/golden/make_resnet50_kat.py      # rewrites resnet50_trunk_kat.json
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def bn(prefix, c, out):
    for n in ('weight', 'bias', 'running_mean', 'running_var'):
        out[prefix + '.' + n] = [c]
    out[prefix + '.num_batches_tracked'] = []


def main():
    keys = {}
    convs = {}       # conv key -> (stride, padding)
    keys['conv1.weight'] = [64, 3, 7, 7]
    convs['conv1'] = (2, 3)
    bn('bn1', 64, keys)
    c_in = 64
    stages = []
    for name, blocks, width, stride in (('layer1', 3, 64, 1), ('layer2', 4, 128, 2), ('layer3', 6, 256, 2)):
        for b in range(blocks):
            p = '%s.%d' % (name, b)
            s = stride if b == 0 else 1
            keys[p + '.conv1.weight'] = [width, c_in, 1, 1]
            convs[p + '.conv1'] = (1, 0)
            bn(p + '.bn1', width, keys)
            keys[p + '.conv2.weight'] = [width, width, 3, 3]
            convs[p + '.conv2'] = (s, 1)
            bn(p + '.bn2', width, keys)
            keys[p + '.conv3.weight'] = [4 * width, width, 1, 1]
            convs[p + '.conv3'] = (1, 0)
            bn(p + '.bn3', 4 * width, keys)
            if b == 0:
                keys[p + '.downsample.0.weight'] = [4 * width, c_in, 1, 1]
                convs[p + '.downsample.0'] = (s, 0)
                bn(p + '.downsample.1', 4 * width, keys)
            c_in = 4 * width
        stages.append(name)

    def count(prefixes):
        n = 0
        for k, shp in keys.items():
            if k.split('.')[0] in prefixes and not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
                m = 1
                for d in shp:
                    m *= d
                n += m
        return n
    params = {'stem': count(('conv1', 'bn1')), 'layer1': count(('layer1',)), 'layer2': count(('layer2',)),
              'layer3': count(('layer3',))}
    assert params == {'stem': 9536, 'layer1': 215808, 'layer2': 1219584, 'layer3': 7098368}, params
    kat = {'source': 'torchvision.models.resnet50 (v1.5), conv1..layer3; stated from the published architecture, see make_resnet50_kat.py',
           'state_dict': keys, 'conv_stride_padding': {k: list(v) for k, v in convs.items()},
           'trainable_parameters': params, 'maxpool': {'kernel': 3, 'stride': 2, 'padding': 1},
           'output_channels': {'layer1': 256, 'layer2': 512, 'layer3': 1024},
           'output_stride': {'stem': 2, 'maxpool': 4, 'layer1': 4, 'layer2': 8, 'layer3': 16}}
    with open(os.path.join(HERE, 'resnet50_trunk_kat.json'), 'w') as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print('resnet50_trunk_kat.json:', len(keys), 'state-dict entries,', sum(params.values()), 'parameters')


if __name__ == '__main__':
    main()
