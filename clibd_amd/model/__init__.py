"""Mirror of the reference package `bioscanclip.model` (same class / function names)."""
from .dna_encoder import CLIBDDNAEncoder, BertForMaskedLM, BertModel, BertConfigLite, load_pre_trained_bioscan_bert, get_sequence_pipeline
from .image_encoder import CLIBDImageEncoder, VisionTransformer, create_vit
from .language_encoder import CLIBDLanguageEncoder, load_pre_trained_bert
from .loss_func import ClipLoss, ContrastiveLoss, construct_label_metrix, gather_features
from .simple_clip import SimpleCLIP, initialize_model_and_load_from_checkpoint, load_clip_model

__all__ = ["CLIBDDNAEncoder", "CLIBDImageEncoder", "CLIBDLanguageEncoder", "SimpleCLIP", "load_clip_model", "initialize_model_and_load_from_checkpoint", "ClipLoss", "ContrastiveLoss",
           "construct_label_metrix", "gather_features", "create_vit", "VisionTransformer", "BertForMaskedLM", "BertModel", "BertConfigLite",
           "load_pre_trained_bioscan_bert", "load_pre_trained_bert", "get_sequence_pipeline"]
