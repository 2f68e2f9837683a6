"""vit_unet - MI355X-native drop-in for the hot path of benayas1/vit-unet."""
__version__ = "0.1.0"
