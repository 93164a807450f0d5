"""MI355X-native MaskCycleGAN-VC hot path (drop-in for the reference's ``mask_cyclegan_vc`` package)."""
