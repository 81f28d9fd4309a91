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


scalogram_resnet_architecture_7 = {'model': ScalogramResidualEncoder, 'phase': True, 'scalogram_offset_zero': False,
                                   'scalogram_output_power': 1., 'scalogram_scaling': 1., 'scalogram_pooling': None,
                                   'blocks': _arch7_blocks(), 'activation_register': None}


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
