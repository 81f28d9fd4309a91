"""MI355X-native CPC-audio train step behind the reference's Python surface.

Modules keep the reference's names (audio_model, contrastive_estimation_training, audio_dataset) so that
``from audio_model import *`` style code can point at this directory instead.  Import via the ``cpc_audio_amd`` alias.
"""
__all__ = ["audio_model", "contrastive_estimation_training", "audio_dataset", "engine", "attention_model", "constant_q_transform",
           "scalogram_model", "scalogram_engine", "configs"]
