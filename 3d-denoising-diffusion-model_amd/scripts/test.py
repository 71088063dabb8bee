"""
Inference entry point with the reference's command line (scripts/test.py:264-278
+ test_DDPM_3d_mpi.sh flags): denoise one whole-body PET volume by tiling it
into sub-volumes, sampling every sub-volume on the GPU(s) and blending the
results with a 3-D Hann window.

    python scripts/test.py --model_path ckpt.pt --base_samples vol.npz --save_dir out \
        --large_size 96 --small_size 96 --num_channels 128 --num_head_channels 64 \
        --attention_resolutions 1000 --learn_sigma True --resblock_updown True \
        --use_scale_shift_norm True --timestep_respacing 250
    python -m torch.distributed.run --nproc-per-node 8 scripts/test.py ...   # one rank per GPU (RCCL)

Differences from the reference script, all on the host side: `.npz`/`.npy`
inputs are accepted besides `.tif` (README.md:67 asks users to edit the loader;
scripts/test.py:187 rejects them); volumes need not be 200x200; `--use_ddim`
is honoured (the reference parses it but always runs DDPM, scripts/test.py:63);
ranks get an evenly padded work list (the reference hangs in all_gather on an
uneven one); `--model_path ""` uses seeded synthetic weights (no checkpoint
ships with the reference).
"""

import argparse
import os
import sys

sys.path.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import numpy as np
import torch as th

from guided_diffusion import dist_util, logger, patches, synth
from guided_diffusion.script_util import (
    add_dict_to_argparser,
    args_to_dict,
    sr_create_model_and_diffusion,
    sr_model_and_diffusion_defaults,
)


def create_argparser():
    defaults = dict(save_dir="", clip_denoised=True, batch_size=1, use_ddim=False, eta=0.0,
                    timestep_respacing="", base_samples="", model_path="",
                    # launch-side extras (not in the reference): collective backend ("" = RCCL on
                    # GPUs) and all ranks on cuda:0, to rehearse the multi-rank flow on a one-GPU box
                    dist_backend="", share_gpu=False,
                    # one captured hipGraph per UNet forward / the library's own launch plan (both bit-identical
                    # to the default launch-by-launch replay of the Python plan)
                    step_graph=False, native_plan=False)
    defaults.update(sr_model_and_diffusion_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


def main(argv=None):
    args = create_argparser().parse_args(argv)
    dist_util.setup_dist(backend=args.dist_backend or None, share_gpu=args.share_gpu)
    logger.configure(dir=args.save_dir)
    dev = dist_util.dev()

    logger.log("creating model...")
    model, diffusion = sr_create_model_and_diffusion(
        **args_to_dict(args, sr_model_and_diffusion_defaults().keys()))
    if args.model_path:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"))
    else:
        logger.log("no --model_path: using seeded synthetic weights")
        model.load_state_dict({k: th.from_numpy(synth.synth_param(k, tuple(v.shape)))
                               for k, v in model.state_dict().items()})
    model.to(dev)
    if args.use_fp16:
        model.convert_to_fp16()
    model.step_graph, model.native_plan = args.step_graph, args.native_plan
    model.eval()

    logger.log("loading data...")
    vol = patches.load_volume(args.base_samples)                 # (D, H, W)
    res = args.large_size
    low_res, grid = patches.split_volume(vol, res)               # (P, 1, Z, H, W)
    logger.log(f"volume {vol.shape}: {len(grid)} patches of {res}^3")

    # Work units are batches of --batch_size patches; batch b goes to rank b mod W
    # (scripts/test.py:243 with batch_size 1), every rank runs the same number of rounds.
    rank = dist_util.rank()
    bs = max(1, args.batch_size)
    n_batches = (len(grid) + bs - 1) // bs
    done = {}
    sample_loop = diffusion.ddim_sample_loop if args.use_ddim else diffusion.p_sample_loop
    extra = dict(eta=args.eta) if args.use_ddim else {}
    for b in dist_util.partition(n_batches):
        block = th.zeros(bs, 1, res, res, res, device=dev)                  # padded so collectives stay aligned
        if b is not None:
            idx = list(range(b * bs, min((b + 1) * bs, len(grid))))
            cond = th.from_numpy(low_res[idx]).to(dev)
            shape = tuple(cond.shape)
            # all randomness keyed by the GLOBAL patch index (one generator per patch): the result
            # depends neither on the world size nor on the batch size
            gens = [dist_util.volume_generator(i, seed=10, device=dev) for i in idx]

            def draw(_k=None, _img=None):
                return th.cat([th.randn(1, *shape[1:], device=dev, generator=g) for g in gens])

            noise = draw()
            logger.log(f"rank {rank}: patches {idx} shape={shape}")
            sample = sample_loop(model, shape, noise, clip_denoised=args.clip_denoised,
                                 model_kwargs={"low_res": cond}, step_noise=draw, **extra)
            block[:len(idx)] = sample.permute(0, 1, 3, 4, 2)                # (B,1,Z,H,W) -> (B,1,H,W,Z)
        for bb, blk in dist_util.gather_round(block, b):
            for j, i in enumerate(range(bb * bs, min((bb + 1) * bs, len(grid)))):
                done[i] = blk[j, 0].cpu().numpy()
    if not done:
        logger.log("No samples were generated. Exiting.")
        return None

    logger.log("Reconstructing full image with Hann window blending...")
    ordered = [done[i] for i in range(len(grid))]
    result, weight = patches.stitch_patches(ordered, grid, vol.shape, res)  # (H, W, Z)
    orig_std, den_std = float(vol.std()), float(result.std())
    logger.log(f"  Original std: {orig_std:.4f}  Denoised std: {den_std:.4f}")

    out_path = None
    if rank == 0:
        base = os.path.basename(args.base_samples)
        for ext in (".tiff", ".tif", ".npz", ".npy"):
            if base.lower().endswith(ext):
                base = base[:-len(ext)]
        out_path = os.path.join(logger.get_dir(), f"denoised_{base}.npz")
        logger.log(f"saving to {out_path}")
        np.savez(out_path, result)                                           # key 'arr_0', (H,W,Z) like the reference
        if args.base_samples.lower().endswith((".tif", ".tiff")):
            from guided_diffusion import tiff_io
            tiff_path = out_path.replace(".npz", ".tif")
            tiff_io.imwrite(tiff_path, result.transpose(2, 0, 1).astype(np.float32))   # (H,W,Z) -> (Z,H,W), no scaling
            logger.log(f"Saved denoised TIFF: {tiff_path}")
    dist_util.barrier()
    logger.log("Full image denoising complete")
    return out_path


if __name__ == "__main__":
    main()
