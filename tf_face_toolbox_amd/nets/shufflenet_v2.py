"""ShuffleNet-v2 on the graph engine -- host-side mirror of nets/shufflenet_v2.py:32-420.

Same classes and constructor arguments (`ShuffleNet_v2_small(alpha, se, residual, ...)`, `_middle`, `_large`),
variable names (`ShuffleNet_v2_small_x2/conv2/resBlock_0/separable_conv2_3x3/{depthwise_weights,pointwise_weights}`,
`.../BatchNorm/*`, `classifier/fc_classifier/weights`) and forward / loss_function / param_list behaviour.

MI355X mapping
  * channel widths (12, 24, 122, 244, 488, 976 ...) are stored rounded up to 64 (GraphNet.channel_pad): the 1x1
    convs -- ~97 % of the FLOPs -- run on the MFMA implicit-GEMM kernels at 5 % padding overhead for the x2 net,
    and every elementwise kernel keeps float4 accesses;
  * tf.concat + _channel_shuffle + the next block's _channel_split collapse into two launches of one table-driven
    channel-gather kernel; the concatenated tensor never exists in HBM;
  * the depthwise 3x3 is its own HBM-bound kernel (9 MAC / element -- nothing for the matrix cores to do).
112x112 input: 112 -> 56 (3x3 s2) -> 28 (max-pool) -> 14 / 7 / 4 over the stages -> 1x1 to 2048 -> GAP."""
from .graph import GraphNet


class ShuffleNet_v2_small(GraphNet):
    channel_pad = 64

    def __init__(self, alpha=1.0, se=False, residual=False, weight_decay=0.0005, data_format='NCHW',
                 name='ShuffleNet_v2_small', seed=0):
        super(ShuffleNet_v2_small, self).__init__(weight_decay, data_format, name, seed)
        if alpha == 0.5:                                           # nets/shufflenet_v2.py:40-50
            self.num_outputs = [24, 48, 96, 1024]
            self.name += '_x0_5'
        elif alpha == 1.0:
            self.num_outputs = [58, 116, 232, 1024]
        elif alpha == 1.5:
            self.num_outputs = [88, 176, 352, 1024]
            self.name += '_x1_5'
        elif alpha == 2.0:
            self.num_outputs = [122, 244, 488, 2048]
            self.name += '_x2'
        else:
            raise AttributeError("'ShuffleNet_v2_small' object has no attribute 'num_outputs'")   # what :157 dies with
        self.se = se
        if se:
            self.name += '_se'
        self.residual = residual
        if residual:
            self.name += '_res'
        self.num_block = [4, 8, 4]                                 # :156,162,168
        self.stem = [('conv_3x3', 24, 2)]                          # :149
        self.feature_name = 'features'

    # -- graph construction ------------------------------------------------------------------------
    def _bn(self, g, spec, scope, out, cout, relu):
        spec.append((scope + '/BatchNorm/gamma', (cout,), 'gamma'))
        spec.append((scope + '/BatchNorm/beta', (cout,), 'beta'))
        g.append(('bn', out + '/bn', out + '/z', scope + '/BatchNorm'))
        if relu:
            g.append(('relu', out, out + '/bn'))
            return out
        return out + '/bn'

    def conv2d(self, g, spec, scope, out, inp, cin, num_outputs, kernel_size, stride=1):
        """layers.conv2d under the arg_scope of :130-135: no bias, batch_norm, ReLU."""
        spec.append((scope + '/weights', (kernel_size, kernel_size, cin, num_outputs), 'conv_w'))
        g.append(('conv', out + '/z', inp, scope + '/weights', stride))
        return self._bn(g, spec, scope, out, num_outputs, True)

    def separable_conv2d(self, g, spec, scope, out, inp, cin, num_outputs, stride):
        """layers.separable_conv2d under :136-143: depthwise 3x3 (multiplier 1) -> pointwise 1x1 -> batch_norm, no
        activation, no bias."""
        spec.append((scope + '/depthwise_weights', (3, 3, cin, 1), 'dw_w'))
        spec.append((scope + '/pointwise_weights', (1, 1, cin, num_outputs), 'conv_w'))
        g.append(('dwconv', out + '/dw', inp, scope + '/depthwise_weights', stride))
        g.append(('conv', out + '/z', out + '/dw', scope + '/pointwise_weights', 1))
        return self._bn(g, spec, scope, out, num_outputs, False)

    def separable_resBlock(self, g, spec, scope, t, halves, widths, num_outputs, stride=1, last=False):
        """nets/shufflenet_v2.py:87-116.  `halves` = the two tensors _channel_split (:93) yields -- produced by the
        previous block's fused concat/shuffle/split -- and `widths` their channel counts.  The residual flag of :91
        compares num_outputs with the UNSPLIT channel count, which is 2*num_outputs for every stride-1 block, so it
        never fires; it is evaluated the same way here."""
        s_in, x_in = halves
        residual_flag = self.residual and (stride == 1 and num_outputs == widths[0] + widths[1])
        assert not residual_flag
        shortcut = s_in
        if stride != 1:
            shortcut = self.separable_conv2d(g, spec, scope + '/separable_conv_shortcut_3x3', t + '/ss', s_in, widths[0], num_outputs, stride)
            shortcut = self.conv2d(g, spec, scope + '/conv_shortcut_1x1', t + '/sc', shortcut, num_outputs, num_outputs, 1)
        x = self.conv2d(g, spec, scope + '/conv1_1x1', t + '/c1', x_in, widths[1], num_outputs, 1)
        x = self.separable_conv2d(g, spec, scope + '/separable_conv2_3x3', t + '/c2', x, num_outputs, num_outputs, stride)
        x = self.conv2d(g, spec, scope + '/conv3_1x1', t + '/c3', x, num_outputs, num_outputs, 1)
        if self.se:                                                # :79-85: two biased 1x1 convs on the squeezed map
            spec.extend([(scope + '/Conv/weights', (1, 1, num_outputs, num_outputs // 2), 'fc_w'),
                         (scope + '/Conv/biases', (num_outputs // 2,), 'bias'),
                         (scope + '/Conv_1/weights', (1, 1, num_outputs // 2, num_outputs), 'fc_w'),
                         (scope + '/Conv_1/biases', (num_outputs,), 'bias')])
            g.append(('se', t + '/se', x, scope, 'Conv', 'Conv_1'))
            x = t + '/se'
        if last:
            g.append(('shufcat', t, shortcut, x, self.data_format))                         # :112-113
            return t
        g.append(('shufsplit', t + '/s', shortcut, x, t + '/x', self.data_format))         # :112-113 + the next :93
        return (t + '/s', t + '/x')

    def stem_graph(self, g, spec, in_ch):
        scope, cout, stride = self.stem[0]
        return self.conv2d(g, spec, '%s/conv1/%s' % (self.name, scope), 'conv1', 'images', in_ch, cout, 3, stride), cout

    def build_graph(self, in_ch, num_classes):
        g, spec = [], []
        x, c = self.stem_graph(g, spec, in_ch)                                              # :148-149
        g.append(('maxpool', 'pool1', x))                                                  # :151
        g.append(('split', 'pool1/s', 'pool1', 'pool1/x'))
        halves, widths = ('pool1/s', 'pool1/x'), (int(0.5 * c), c - int(0.5 * c))
        nstage = len(self.num_block)
        for si, nb in enumerate(self.num_block):                                           # :155-172
            scope = 'conv%d' % (si + 2)
            for idx in range(nb):
                last = si == nstage - 1 and idx == nb - 1
                halves = self.separable_resBlock(g, spec, '%s/%s/resBlock_%d' % (self.name, scope, idx), '%sb%d' % (scope, idx),
                                                 halves, widths, self.num_outputs[si], 2 if not idx else 1, last)
                widths = (self.num_outputs[si], self.num_outputs[si])
        x = self.conv2d(g, spec, self.name + '/conv5/conv_1x1', 'conv_last', halves, 2 * self.num_outputs[nstage - 1],
                        self.num_outputs[nstage], 1)                                       # :174-177
        g.append(('gap', 'features', x))                                                   # :180
        g.append(('dropout', 'features_drop', 'features', 0.5))                            # :191
        g.append(('fc', 'logits', 'features_drop', 'classifier/fc_classifier/weights', None))   # :192-196
        spec.append(('classifier/fc_classifier/weights', (self.num_outputs[nstage], num_classes), 'cls_w'))
        return g, spec


class ShuffleNet_v2_middle(ShuffleNet_v2_small):
    def __init__(self, se=False, residual=False, weight_decay=0.0005, data_format='NCHW', name='ShuffleNet_v2_middle', seed=0):
        super(ShuffleNet_v2_middle, self).__init__(1.0, se, residual, weight_decay, data_format, name, seed)
        self.num_outputs = [244, 488, 976, 1952, 2048]             # nets/shufflenet_v2.py:234
        self.num_block = [3, 4, 6, 3]                              # :265,271,277,283
        self.stem = [('conv_3x3', 64, 2)]                          # :260


class ShuffleNet_v2_large(ShuffleNet_v2_small):
    """nets/shufflenet_v2.py:299-379.  se=True, residual=True (:305).  The stem applies conv1_3x3, conv2_3x3 and
    conv3_3x3 all to `inputs` (:336-340): only conv3_3x3 (stride 1, 128 channels, on the full 112x112 image) reaches
    the max-pool; the first two are variables that only ever see weight decay."""

    def __init__(self, weight_decay=0.0005, data_format='NCHW', name='ShuffleNet_v2_large', seed=0):
        super(ShuffleNet_v2_large, self).__init__(1.0, True, True, weight_decay, data_format, name, seed)
        self.num_outputs = [340, 680, 1360, 2720, 2048]            # :306
        self.num_block = [10, 10, 23, 10]                          # :346,352,358,364

    def stem_graph(self, g, spec, in_ch):
        for scope in ('conv1_3x3', 'conv2_3x3'):                   # dead branches (:336-339)
            sc = '%s/conv1/%s' % (self.name, scope)
            spec.append((sc + '/weights', (3, 3, in_ch, 64), 'conv_w'))
            spec.append((sc + '/BatchNorm/gamma', (64,), 'gamma'))
            spec.append((sc + '/BatchNorm/beta', (64,), 'beta'))
        return self.conv2d(g, spec, self.name + '/conv1/conv3_3x3', 'conv1', 'images', in_ch, 128, 3, 1), 128
