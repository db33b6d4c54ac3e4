"""Net factory + Network ABC: the host-side mirror of the reference's nets/net_base.py.

Same names, arguments and error behaviour as nets/net_base.py:22-63 (`net_select`) and
:65-101 (`Network`).  What was graph construction in TF1 is eager here: `forward` /
`loss_function` enqueue HIP kernels, and because there is no `tf.gradients`
(data_parallel.py:33) a Network also implements `backward()`, which fills the flat
gradient arena the parallel wrappers all-reduce.
"""
import abc


def net_select(name, data_format='NCHW', weight_decay=5e-4):
    """nets/net_base.py:22-63.  Names kept verbatim; `SphereNet-ASoftmax` is the margin net the
    reference's `DataParallel_margin` (data_parallel.py:220) expects but whose code is missing
    from the snapshot (README.md:14,19)."""
    if name == 'SphereNet':
        from .sphere import SphereNet
        network = SphereNet(data_format=data_format, weight_decay=weight_decay)
    elif name == 'SphereNet-ASoftmax':
        from .sphere import SphereNetMargin
        network = SphereNetMargin(data_format=data_format, weight_decay=weight_decay)
    elif name == 'ResNet-50':
        from .resnet import ResNet
        network = ResNet(num_layers=50, data_format=data_format, weight_decay=weight_decay)
    elif name == 'ResNet-26':                    # not a reference factory name; the class accepts 26 (nets/resnet.py:39-40)
        from .resnet import ResNet
        network = ResNet(num_layers=26, data_format=data_format, weight_decay=weight_decay)
    elif name in ('ResNeXt-26', 'ResNeXt-50'):   # nets/net_base.py:27-31 exposes -26; -50 is BASELINE config 3
        from .resnet import ResNeXt
        network = ResNeXt(num_layers=int(name.split('-')[1]), num_card=32, data_format=data_format, weight_decay=weight_decay)
    elif name in ('ResNeXt-50-center', 'ResNeXt-26-center'):      # config 3: + center loss on the pooled features (loss.py:29-45)
        from .resnet import ResNeXt
        network = ResNeXt(num_layers=int(name.split('-')[1]), num_card=32, data_format=data_format, weight_decay=weight_decay,
                          head='softmax+center', center_weight=0.008)
    elif name == 'SENet-50':
        from .resnet import SENet
        network = SENet(num_layers=50, data_format=data_format, weight_decay=weight_decay)
    elif name == 'SENet-50-triplet':             # config 4: batch-hard triplet (loss.py:47-78), no classifier
        from .resnet import SENet
        network = SENet(num_layers=50, data_format=data_format, weight_decay=weight_decay, head='triplet')
    elif name == 'ShuffleNet-v2-small':                                  # nets/net_base.py:37-42
        from .shufflenet_v2 import ShuffleNet_v2_small
        network = ShuffleNet_v2_small(alpha=2.0, se=False, residual=False, data_format=data_format, weight_decay=weight_decay)
    elif name == 'ShuffleNet-v2-middle':                                 # :43-47
        from .shufflenet_v2 import ShuffleNet_v2_middle
        network = ShuffleNet_v2_middle(se=False, residual=False, data_format=data_format, weight_decay=weight_decay)
    elif name == 'ShuffleNet-v2-large':                                  # :48-51
        from .shufflenet_v2 import ShuffleNet_v2_large
        network = ShuffleNet_v2_large(data_format=data_format, weight_decay=weight_decay)
    elif name in ('MobileNet-v2', 'Inception-v4', 'VGG16', 'AlexNet'):
        # nets/net_base.py:52-59 `pass` branches: the reference dies with UnboundLocalError here
        raise UnboundLocalError("local variable 'network' referenced before assignment")
    else:
        raise ValueError('Unsupport network architecture.')
    return network


class Network(abc.ABC):
    """nets/net_base.py:65-101."""

    needs_labels = False       # True for margin nets: forward(images, labels, num_classes=...)

    def __init__(self, weight_decay, data_format, name=None):
        assert data_format in ['NCHW', 'NHWC'], 'Unknown data format.'
        self.data_format = data_format
        self.channel_axis = 1 if self.data_format == 'NCHW' else 3
        self.spatial_axis = [2, 3] if self.data_format == 'NCHW' else [1, 2]
        self.weight_decay = weight_decay
        self.name = name

    @abc.abstractmethod
    def backbone(self, inputs, is_training, reuse):
        pass

    @abc.abstractmethod
    def forward(self, images, num_classes, is_training):
        pass

    @abc.abstractmethod
    def loss_function(self, scope, labels, **logits):
        pass

    @abc.abstractmethod
    def backward(self):
        """Replaces tf.gradients(total_loss, params): fill the gradient arena for the
        batch that the last forward()/loss_function() saw."""

    def param_list(self, is_training, trainable, scope=None):
        raise NotImplementedError

    def mult_lr_list(self, scope=None):
        return [1.0 for _ in self.param_list(is_training=True, trainable=True, scope=scope)]


_SIDE_STREAMS = {}


def side_stream(device, priority=0):
    """The process's second stream on `device` (filter gradients beside the data-gradient chain, the forward walk's second half shard, the
    BN nets' side work): ONE per (device, priority), shared by every net of the process.  A stream per net made the walk's overlap a
    matter of luck: the runtime maps streams onto a few hardware queues round-robin, and the n-th stream created can share its queue
    with the current stream -- the 64-image SphereNet step then read 6.5 ms instead of 5.9 in bench.py's `other_configs` leg, depending
    on which nets had been built before it."""
    import torch
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(priority))
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=int(priority))
    return s
