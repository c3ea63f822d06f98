"""open-pandora_amd: MI355X-native (gfx950) implementation of Open-Pandora's DDIM / 3-D U-Net
denoising hot path.  Import as `open_pandora_amd` (the shim package next to this directory)."""
__version__ = "0.1.0"
