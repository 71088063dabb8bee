"""
MI355X-native 3D DDPM denoising sampler behind the reference's
``guided_diffusion`` Python surface (script_util / gaussian_diffusion /
respace / unet).  Compute runs in hand-written HIP kernels (csrc/) reached
through the C-ABI in include/ddpm3d.h; there is no CPU or eager fallback.
"""
