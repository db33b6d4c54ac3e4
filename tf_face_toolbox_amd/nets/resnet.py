"""ResNet (bottleneck, conv-BN-ReLU) on the graph engine -- host-side mirror of nets/resnet.py:24-193.

Same constructor arguments, variable names (`ResNet-50/conv2/resBlock_0/conv1_1x1/weights`,
`.../BatchNorm/{gamma,beta,moving_mean,moving_variance}`, `classifier/fc_classifier/weights`) and
forward / loss_function / param_list / pretrained_param behaviour.  112x112 input: 112 -> 56 (7x7 s2) ->
28 (max-pool) -> 28 / 14 / 7 / 4 over the four stages -> GAP -> dropout 0.5 -> classifier."""
from .graph import GraphNet


class ResNet(GraphNet):
    def __init__(self, num_layers, pre_act=False, weight_decay=0.0005, data_format='NCHW', name='ResNet', seed=0):
        assert (num_layers - 2) % 3 == 0, "num_layers-2 must be divided by 3."        # nets/resnet.py:31
        self.num_layers = num_layers
        self.pre_act = pre_act
        if pre_act:
            raise NotImplementedError('pre_act=True cannot run in the reference either (nets/resnet.py:80 calls an '
                                      'undefined batch_norm); not built.')
        if self.num_layers in [50, 101]:
            self.num_block = [3, 4, (self.num_layers - 32) // 3, 3]                   # nets/resnet.py:35-36
        elif self.num_layers == 152:
            self.num_block = [3, 8, 36, 3]
        elif self.num_layers == 26:
            self.num_block = [2, 2, 2, 2]
        else:
            raise ValueError('Unsupported num_layers.')
        self.num_outputs = [256, 512, 1024, 2048]
        super(ResNet, self).__init__(weight_decay, data_format, name + '-' + str(num_layers), seed)
        self.feature_name = 'features'

    # -- graph construction (what backbone()/forward() build as TF ops in the reference) -----------
    def conv_bn_relu(self, g, spec, scope, out, inp, cin, num_outputs, kernel_size, stride=1, relu=True):
        """nets/resnet.py:47-61: conv (no bias) -> batch_norm -> optional ReLU; returns the output tensor name."""
        spec.append((scope + '/weights', (kernel_size, kernel_size, cin, num_outputs), 'conv_w'))
        spec.append((scope + '/BatchNorm/gamma', (num_outputs,), 'gamma'))
        spec.append((scope + '/BatchNorm/beta', (num_outputs,), 'beta'))
        g.append(('conv', out + '/z', inp, scope + '/weights', stride))
        g.append(('bn', out + '/bn', out + '/z', scope + '/BatchNorm'))
        if relu:
            g.append(('relu', out, out + '/bn'))
            return out
        return out + '/bn'

    def resBlock(self, g, spec, scope, t, x, cin, num_outputs, stride=1):
        """nets/resnet.py:63-92."""
        shortcut = x
        if stride != 1 or cin != num_outputs:
            shortcut = self.conv_bn_relu(g, spec, scope + '/conv_shortcut_1x1', t + '/sc', x, cin, num_outputs, 1, stride, relu=False)
        y = self.conv_bn_relu(g, spec, scope + '/conv1_1x1', t + '/c1', x, cin, num_outputs // 4, 1, 1)
        y = self.conv_bn_relu(g, spec, scope + '/conv2_3x3', t + '/c2', y, num_outputs // 4, num_outputs // 4, 3, stride)
        y = self.conv_bn_relu(g, spec, scope + '/conv3_1x1', t + '/c3', y, num_outputs // 4, num_outputs, 1, 1, relu=False)
        g.append(('add', t + '/sum', y, shortcut))
        g.append(('relu', t, t + '/sum'))
        return t

    def build_graph(self, in_ch, num_classes):
        g, spec = [], []
        x = self.conv_bn_relu(g, spec, self.name + '/conv1/conv_7x7', 'conv1', 'images', in_ch, 64, 7, 2)     # :109-113
        g.append(('maxpool', 'pool1', x))                                                                   # :115
        x, cin = 'pool1', 64
        for si, nb in enumerate(self.num_block):                                                            # :119-140
            for idx in range(nb):
                stride = 2 if (idx == 0 and si > 0) else 1
                x = self.resBlock(g, spec, '%s/conv%d/resBlock_%d' % (self.name, si + 2, idx), 's%db%d' % (si + 2, idx),
                                  x, cin, self.num_outputs[si], stride)
                cin = self.num_outputs[si]
        g.append(('gap', 'features', x))                                                                    # :142
        if getattr(self, 'head', 'softmax') == 'triplet':      # metric-learning head on the pooled features: no classifier
            return g, spec
        g.append(('dropout', 'features_drop', 'features', 0.5))                                             # :152
        g.append(('fc', 'logits', 'features_drop', 'classifier/fc_classifier/weights', None))               # :153-157
        spec.append(('classifier/fc_classifier/weights', (self.num_outputs[3], num_classes), 'cls_w'))
        return g, spec


class ResNeXt(ResNet):
    """nets/resnext.py:21-67 as INTENDED: bottleneck with mid = C/2 channels, the 3x3 conv grouped x num_card (one
    grouped-conv kernel instead of tf.split + 32 convs + tf.concat), conv-BN-ReLU like its base class.  The snapshot's
    override passes a kwarg its helper does not take and drops BN/ReLU; that cannot run, so the
    parity target is the build's own oracle.  The 32 per-group variables `conv2_3x3_group_<i>/weights` are one stacked
    variable `conv2_3x3/weights` of shape [32,3,3,gw,gw]."""

    def __init__(self, num_layers, num_card=32, weight_decay=0.0005, data_format='NCHW', name='ResNeXt', seed=0,
                 head='softmax', center_weight=0.0):
        self.num_card = num_card
        super(ResNeXt, self).__init__(num_layers, weight_decay=weight_decay, data_format=data_format, name=name, seed=seed)
        self.head = head
        self.center_weight = center_weight

    def resBlock(self, g, spec, scope, t, x, cin, num_outputs, stride=1):
        assert num_outputs % 2 == 0, "num_outputs must be divided by 2."                   # nets/resnext.py:53
        shortcut = x
        if stride != 1 or cin != num_outputs:
            shortcut = self.conv_bn_relu(g, spec, scope + '/conv_1x1_shortcut', t + '/sc', x, cin, num_outputs, 1, stride, relu=False)
        mid = num_outputs // 2
        y = self.conv_bn_relu(g, spec, scope + '/conv1_1x1', t + '/c1', x, cin, mid, 1, 1)
        gw = mid // self.num_card
        sc2 = scope + '/conv2_3x3'
        spec.append((sc2 + '/weights', (self.num_card, 3, 3, gw, gw), 'gconv_w'))
        spec.append((sc2 + '/BatchNorm/gamma', (mid,), 'gamma'))
        spec.append((sc2 + '/BatchNorm/beta', (mid,), 'beta'))
        g.append(('gconv', t + '/c2/z', y, sc2 + '/weights', stride, self.num_card))
        g.append(('bn', t + '/c2/bn', t + '/c2/z', sc2 + '/BatchNorm'))
        g.append(('relu', t + '/c2', t + '/c2/bn'))
        y = self.conv_bn_relu(g, spec, scope + '/conv3_1x1', t + '/c3', t + '/c2', mid, num_outputs, 1, 1, relu=False)
        g.append(('add', t + '/sum', y, shortcut))
        g.append(('relu', t, t + '/sum'))
        return t


class SENet(ResNet):
    """SE-ResNet ("SENet-50", BASELINE.json config 4): the ResNet bottleneck of nets/resnet.py:63-92 with the
    squeeze-excitation gate of nets/shufflenet_v2.py:79-85 (GAP -> 1x1 to C/2 + ReLU -> 1x1 to C + sigmoid ->
    channel scale, both with biases, no BN) applied to the block output before the residual add.  The reference has
    no such class: the composition and its oracle are the build's."""

    def __init__(self, num_layers, weight_decay=0.0005, data_format='NCHW', name='SENet', seed=0, head='softmax',
                 triplet_margin=None):
        super(SENet, self).__init__(num_layers, weight_decay=weight_decay, data_format=data_format, name=name, seed=seed)
        self.head = head
        self.triplet_margin = triplet_margin

    def resBlock(self, g, spec, scope, t, x, cin, num_outputs, stride=1):
        shortcut = x
        if stride != 1 or cin != num_outputs:
            shortcut = self.conv_bn_relu(g, spec, scope + '/conv_shortcut_1x1', t + '/sc', x, cin, num_outputs, 1, stride, relu=False)
        y = self.conv_bn_relu(g, spec, scope + '/conv1_1x1', t + '/c1', x, cin, num_outputs // 4, 1, 1)
        y = self.conv_bn_relu(g, spec, scope + '/conv2_3x3', t + '/c2', y, num_outputs // 4, num_outputs // 4, 3, stride)
        y = self.conv_bn_relu(g, spec, scope + '/conv3_1x1', t + '/c3', y, num_outputs // 4, num_outputs, 1, 1, relu=False)
        pre = scope + '/se'
        spec.extend([(pre + '/fc1/weights', (num_outputs, num_outputs // 2), 'fc_w'), (pre + '/fc1/biases', (num_outputs // 2,), 'bias'),
                     (pre + '/fc2/weights', (num_outputs // 2, num_outputs), 'fc_w'), (pre + '/fc2/biases', (num_outputs,), 'bias')])
        g.append(('se', t + '/se', y, pre))
        g.append(('add', t + '/sum', t + '/se', shortcut))
        g.append(('relu', t, t + '/sum'))
        return t
