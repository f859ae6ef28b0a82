"""difashion_amd -- MI355X (gfx950) native implementation of DiFashion's parallel conditional
denoising U-Net path, behind the diffusers UNet2DConditionModel / DDIMScheduler call signatures.

Compute lives in ``csrc/libdifashion_hip.so`` (C ABI: ``include/difashion_hip.h``); this package is
the Python host side mirroring the reference interface.  Importing the package does not need a GPU;
running any op does, and fails loudly if the library was not built (no CPU fallback).
"""
from . import _lib, data, evalio, prompts
from ._lib import DfhError
from .mutual import MutualEncoder
from .pipeline import OutfitSampler, guidance_plan, sample_outfits, sampling_tables, train_forward, training_tables
from .schedulers import DDIMScheduler, PNDMScheduler
from .training import EMAModel, FusedAdamW, clip_grad_norm_, train_step
from .unet import UNet2DConditionModel, UNet2DConditionOutput
from .vae import AutoencoderKL
from .clip import CLIPTextModel
from .difashion import DiFashion

__all__ = [
    "DfhError", "UNet2DConditionModel", "UNet2DConditionOutput", "DDIMScheduler", "PNDMScheduler",
    "MutualEncoder", "OutfitSampler", "sample_outfits", "train_forward", "guidance_plan", "sampling_tables", "training_tables",
    "FusedAdamW", "EMAModel", "clip_grad_norm_", "train_step", "AutoencoderKL", "CLIPTextModel", "DiFashion",
]
