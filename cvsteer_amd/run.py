"""cvsteer-run for MI355X: the reference's batch driver (example/steer.cpp:59-173) over the HIP engine.

    python -m cvsteer_amd.run --input <image | list.txt> --output <dir> [--gain G]
    python -m torch.distributed.run --nproc-per-node 8 -m cvsteer_amd.run --input list.txt --output out

Per image, exactly the reference's per-file body (steer.cpp:69-124): gray f32 (unscaled 0..255) ->
SteerableFiltersG2(gray, 4, 0.67) -> steer at the dominant orientation -> findEdges / findDarkLines /
findBrightLines on the magnitude -> 8-bit via normalize(0,255,MINMAX) or convertTo(gain) ->
<base>_edges.png, <base>_lines_dark.png, <base>_lines_bright.png.  All arithmetic, including the
8-bit conversion, runs on the GPU; only file decoding/encoding is host work (Pillow / numpy, since
OpenCV's imgcodecs are not available).

The reference parallelises over files with cv::parallel_for_ (steer.cpp:169); here the file list is
sharded over the ranks of a torch.distributed job (one process per GPU, contiguous blocks) and each
rank walks its block -- no collective is needed on the data path.

Differences from the reference, on purpose: `--gain` is honoured (the reference passes `--verbose`
as the gain, steer.cpp:167-168); single-channel inputs work (the reference leaves `gray` empty for
them, steer.cpp:79-82); unreadable files are reported, not silently skipped.
"""
import argparse
import os
import sys

import numpy as np

from .batch import shard_range


def read_gray(path):
    """image file -> 2-D array (uint8 or float32), the caller-side imread + BGR2GRAY of steer.cpp:73-82"""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        a = np.load(path)
    else:
        from PIL import Image
        a = np.asarray(Image.open(path).convert("L"))
    if a.ndim == 3:  # H x W x C -> luma, ITU-R 601 like cv::COLOR_BGR2GRAY (channels assumed RGB)
        a = a[..., :3].astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)
    if a.ndim != 2:
        raise ValueError("%s: expected a 2-D (gray) or 3-D (colour) image" % path)
    return a


def write_u8(path, u8):
    if path.lower().endswith(".npy"):
        np.save(path, u8)
    else:
        from PIL import Image
        Image.fromarray(u8).save(path)


def input_list(arg):
    """steer.cpp:156-165: a .txt file (or a name without extension) is a list of files, else one image"""
    if arg.endswith(".txt") or "." not in os.path.basename(arg):
        with open(arg) as f:
            return [ln.strip() for ln in f if ln.strip()]
    return [arg]


def process_file(engine, path, outdir, gain, ext=".png"):
    import torch
    gray = read_gray(path)
    dev = torch.device("cuda", engine.device)
    img = torch.from_numpy(np.ascontiguousarray(gray))
    if img.dtype != torch.uint8:
        img = img.to(torch.float32)
    img = img.to(dev)  # 8-bit images cross PCIe as bytes and are widened by the engine (CVS_DEPTH_U8)
    # one launch; only the three feature maps leave the kernel (set_persist(False): no state planes)
    feat = [torch.empty(tuple(img.shape), dtype=torch.float32, device=dev) for _ in range(3)]
    outs = engine.pipeline(img, out=[None] * 5 + feat)
    base = os.path.splitext(os.path.basename(path))[0]
    written = []
    for plane, suffix in zip(outs[5:], ("_edges", "_lines_dark", "_lines_bright")):
        u8 = engine.convert_u8(plane, gain) if gain > 0 else engine.normalize_u8(plane)
        if outdir:
            dst = os.path.join(outdir, base + suffix + ext)
            write_u8(dst, u8.cpu().numpy())
            written.append(dst)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cvsteer-run", description=__doc__.split("\n")[0])
    ap.add_argument("--input", required=True, help="image file, or a .txt list of image files")
    ap.add_argument("--output", default="", help="output directory")
    ap.add_argument("--gain", type=float, default=0.0, help="gain for the 8-bit output (0 = min-max normalise)")
    ap.add_argument("--ext", default=".png", help="output file extension (.png, .pgm, .npy ...)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)

    import torch
    from . import SteerableFiltersG2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("cvsteer-run needs a HIP device (there is no CPU fallback)")
    files = input_list(args.input)
    lo, hi = shard_range(len(files), world, rank)
    if args.output:
        os.makedirs(args.output, exist_ok=True)
    engine = SteerableFiltersG2(None, 4, 0.67, device=local_rank)
    engine.set_persist(False)  # the driver never revisits an image's basis planes
    failed = 0
    for path in files[lo:hi]:
        try:
            written = process_file(engine, path, args.output, args.gain, args.ext)
            if args.verbose:
                print("[rank %d] %s -> %s" % (rank, path, ", ".join(written) or "(not written)"), flush=True)
        except Exception as exc:  # the reference `continue`s silently on unreadable files
            failed += 1
            print("[rank %d] %s: %s" % (rank, path, exc), file=sys.stderr, flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
