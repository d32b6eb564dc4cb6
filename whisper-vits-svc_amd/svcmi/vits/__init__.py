"""VITS prior encoder, reverse flow and NSF-BigVGAN generator on the svcmi kernels (models.py), parameter table (spec.py)."""
