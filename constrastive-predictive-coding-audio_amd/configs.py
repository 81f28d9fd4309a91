"""Preset configuration dictionaries with the reference's names and values (reference configs/*.py), for the presets the
HIP path covers.  Plain data: 'model' entries point at this package's classes.  Values are the EFFECTIVE ones after the
reference's config modules have been imported (its dict.copy() aliasing shares block dictionaries between architectures:
scalogram_resnet_architecture_7 ends up with batch norm on blocks 0-2 only — SURVEY.md 8a10)."""
import copy

from .audio_model import ConvolutionalArModel
from .attention_model import AttentionModel
from .scalogram_model import ScalogramResidualEncoder, cqt_default_dict  # noqa: F401

# ---- autoregressive context networks (reference configs/autoregressive_model_configs.py:5-127)
ar_conv_default_dict = {'model': ConvolutionalArModel, 'kernel_sizes': [9, 9, 9], 'channel_count': [256, 256, 256, 256],
                        'stride': [1, 1, 1], 'pooling': [1, 2, 2], 'bias': True, 'batch_norm': False, 'residual': False,
                        'encoding_size': 256, 'ar_code_size': 256, 'activation_register': None, 'self_attention': [False] * 3}
ar_conv_architecture_1 = dict(ar_conv_default_dict, channel_count=[256, 512, 512, 256])
ar_conv_architecture_2 = dict(ar_conv_default_dict, kernel_sizes=[5] * 6, channel_count=[256, 512, 512, 256, 256, 256, 256],
                              stride=[1] * 6, pooling=[1, 1, 2, 1, 2, 1], batch_norm=True, residual=True, self_attention=[False] * 6)
ar_conv_architecture_3 = dict(ar_conv_architecture_2, channel_count=[512, 512, 512, 256, 256, 256, 256], encoding_size=512)
ar_conv_architecture_4 = dict(ar_conv_architecture_3, channel_count=[512, 1024, 512, 512, 256, 256, 256])
ar_conv_architecture_5 = dict(ar_conv_architecture_4, kernel_sizes=[5, 4, 3, 3, 3, 5])
# ('self_attention' is read by nothing in the reference's ConvolutionalArModel, audio_model.py:139-161: the entries stay data)
ar_conv_architecture_6 = dict(ar_conv_architecture_5, channel_count=[512] * 5 + [256] * 5, kernel_sizes=[5, 4, 1, 3, 3, 1, 3, 1, 5],
                              pooling=[1, 1, 2, 1, 1, 2, 1, 1, 1], stride=[1] * 9,
                              self_attention=[False, True, False, False, True, False, True, False, False])

attention_default_dict = {'model': AttentionModel, 'channels': 512, 'output_size': 512, 'num_layers': 2, 'num_heads': 8,
                          'feedforward_size': 512, 'sequence_length': 60, 'dropout': 0.1, 'encoding_size': 512, 'ar_code_size': 512}
attention_architecture_1 = dict(attention_default_dict, output_size=256, num_layers=3, ar_code_size=256)
attention_architecture_2 = dict(attention_default_dict, output_size=256, num_layers=6, feedforward_size=2048, ar_code_size=256)

# ---- scalogram encoder (reference configs/scalogram_resnet_configs.py:3-257)
scalogram_block_default_dict = {'in_channels': 64, 'hidden_channels': None, 'out_channels': 64, 'kernel_size_1': (3, 3),
                                'kernel_size_2': (3, 3), 'top_padding_1': None, 'top_padding_2': None, 'padding_1': 0, 'padding_2': 0,
                                'stride_1': 1, 'stride_2': 1, 'pooling_1': 1, 'pooling_2': 1, 'bias': True, 'separable': False,
                                'residual': True, 'batch_norm': False, 'ceil_pooling': False}


def _arch7_blocks():
    b = scalogram_block_default_dict
    return [dict(b, in_channels=1, out_channels=32, stride_1=2, kernel_size_2=(64, 1), top_padding_2=63, batch_norm=True),
            dict(b, in_channels=32, out_channels=128, stride_1=2, kernel_size_2=(30, 1), batch_norm=True),
            dict(b, in_channels=128, out_channels=256, stride_1=2, kernel_size_2=(15, 1), batch_norm=True),
            dict(b, in_channels=256, out_channels=512, kernel_size_1=(2, 2), kernel_size_2=(1, 1), batch_norm=False)]


scalogram_resnet_default_dict = {'model': ScalogramResidualEncoder, 'phase': True, 'scalogram_offset_zero': False,
                                 'scalogram_output_power': 1., 'scalogram_scaling': 1., 'scalogram_pooling': None,
                                 'blocks': [scalogram_block_default_dict] * 3, 'activation_register': None}
scalogram_resnet_architecture_7 = dict(scalogram_resnet_default_dict, blocks=_arch7_blocks())
# (architectures 5 and 6 share architecture 7's block dictionaries in the reference and so equal it after import)
scalogram_resnet_architecture_5 = scalogram_resnet_architecture_6 = scalogram_resnet_architecture_7


def _blk(**kw):
    return dict(scalogram_block_default_dict, **kw)


# The architectures WITHOUT BatchNorm (reference configs/scalogram_resnet_configs.py:46-214; experiments e0-e12 incl. the
# experiments' default dict use 1-4).  padding_1 = padding_2 = 1 is the reference's block_3x3, stride_1 = 2 its *_strided form.
scalogram_resnet_architecture_1 = dict(scalogram_resnet_default_dict, blocks=[
    _blk(in_channels=1, out_channels=32, padding_1=1, padding_2=1),
    _blk(in_channels=32, out_channels=64, padding_1=1, stride_1=2, kernel_size_2=(64, 1), top_padding_2=63),
    _blk(in_channels=64, out_channels=64, padding_1=1),
    _blk(in_channels=64, out_channels=128, padding_1=1, stride_1=2, kernel_size_2=(30, 1)),
    _blk(in_channels=128, out_channels=128, padding_1=1, padding_2=1),
    _blk(in_channels=128, out_channels=256, padding_1=1, stride_1=2, kernel_size_2=(15, 1)),
    _blk(in_channels=256, out_channels=256, padding_1=1)])
# architecture_2_wo_res switches the residual branches of the SHARED block dictionaries off (:132-134), so after import
# architecture_2 has none either
scalogram_resnet_architecture_2 = dict(scalogram_resnet_default_dict, blocks=[
    _blk(in_channels=1, out_channels=32, padding_1=1, stride_1=2, kernel_size_2=(64, 1), top_padding_2=63, residual=False),
    _blk(in_channels=32, out_channels=64, padding_1=1, stride_1=2, kernel_size_2=(30, 1), residual=False),
    _blk(in_channels=64, out_channels=128, padding_1=1, stride_1=2, kernel_size_2=(15, 1), residual=False),
    _blk(in_channels=128, out_channels=256, padding_1=1, residual=False)])
scalogram_resnet_architecture_2_wo_res = scalogram_resnet_architecture_2
scalogram_resnet_architecture_3 = dict(scalogram_resnet_default_dict, blocks=[
    _blk(in_channels=1, out_channels=32, stride_1=2, kernel_size_2=(64, 1), top_padding_2=63),
    _blk(in_channels=32, out_channels=64, stride_1=2, kernel_size_2=(30, 1)),
    _blk(in_channels=64, out_channels=128, stride_1=2, kernel_size_2=(15, 1)),
    _blk(in_channels=128, out_channels=256, kernel_size_1=(2, 2), kernel_size_2=(1, 1))])
scalogram_resnet_architecture_4 = dict(scalogram_resnet_default_dict, blocks=[
    _blk(in_channels=1, out_channels=32, stride_1=2),
    _blk(in_channels=32, out_channels=64, kernel_size_2=(64, 1), top_padding_2=63),
    _blk(in_channels=64, out_channels=128, stride_1=2),
    _blk(in_channels=128, out_channels=128, kernel_size_2=(20, 1)),
    _blk(in_channels=128, out_channels=256, stride_1=2),
    _blk(in_channels=256, out_channels=256, kernel_size_2=(14, 1)),
    _blk(in_channels=256, out_channels=256, kernel_size_1=(1, 3), kernel_size_2=(1, 3))])


# architectures 8 / 9 (:260-318; the high-resolution scalogram of experiments e27-e32, the script default e29 among them):
# power scalogram pooled over two frames, tall FIRST kernels, stride in the second convolution, BatchNorm everywhere
def _arch8_blocks():
    bn = dict(padding_1=1, padding_2=1, batch_norm=True)
    return [_blk(in_channels=1, out_channels=32, kernel_size_1=(65, 1), stride_2=2, **dict(bn, padding_1=0)),
            _blk(in_channels=32, out_channels=32, **bn),
            _blk(in_channels=32, out_channels=64, kernel_size_1=(33, 1), stride_2=2, **dict(bn, padding_1=0)),
            _blk(in_channels=64, out_channels=64, **bn),
            _blk(in_channels=64, out_channels=128, kernel_size_1=(16, 1), stride_2=2, **dict(bn, padding_1=0)),
            _blk(in_channels=128, out_channels=128, **bn),
            _blk(in_channels=128, out_channels=256, kernel_size_1=(9, 1), **dict(bn, padding_1=0)),
            _blk(in_channels=256, out_channels=512, batch_norm=True)]


scalogram_resnet_architecture_8 = dict(scalogram_resnet_default_dict, phase=False, scalogram_offset_zero=True, scalogram_output_power=2.,
                                       scalogram_pooling=[1, 2], blocks=_arch8_blocks())
scalogram_resnet_architecture_9 = dict(scalogram_resnet_architecture_8, scalogram_scaling=10.)
cqt_high_res_dict = dict(cqt_default_dict, sample_rate=44100, n_bins=292, hop_length=256)


# ---- ScalogramResidualEncoder as the context network (reference configs/autoregressive_model_configs.py:66-102; effective
# values: architecture_2's edits of the shared block dictionaries also reach architecture_1 there — both are given here as
# the reference's module leaves them after import)
def _ar_resnet_blocks(first_in, last_out):
    b = dict(scalogram_block_default_dict, in_channels=256, out_channels=512, kernel_size_1=(1, 9), kernel_size_2=(1, 1),
             ceil_pooling=True, pooling_1=2, batch_norm=True)
    return [dict(b, in_channels=first_in), dict(b, in_channels=512), dict(b, in_channels=512, out_channels=last_out, kernel_size_1=(1, 8))]


ar_resnet_architecture_2 = {'model': ScalogramResidualEncoder, 'phase': False, 'blocks': _ar_resnet_blocks(512, 256),
                            'encoding_size': 512, 'ar_code_size': 256, 'activation_register': None}


def fresh(config):
    """A deep copy of a preset: the model constructors write into the dictionaries they are given (hidden_channels,
    the first block's in_channels with phase=True), exactly as the reference's do."""
    return copy.deepcopy(config)
