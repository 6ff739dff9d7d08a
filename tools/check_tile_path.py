#!/usr/bin/env python3
"""The window tiles of crf_decode_pipelined share a 64-VGPR kernel with the Viterbi workgroups, whose SGPR spills take one
of the 64: a change anywhere in the kernel can push a tile value into scratch (44 instead of 35 us per step, measured).
This compiles crf_kernels.hip to assembly and fails when the tile path of the kernel holds a scratch access beyond the
entry spill / reload of v0 (one store, one load).  usage: tools/check_tile_path.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "gecco_amd", "csrc", "crf_kernels.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    # the one-tile shape (batches that do not fill the chip) is compiled for seven workgroups per CU: no scratch at all
    meta = "\n".join(lines)
    meta = meta[meta.index("amdhsa.kernels:"):]
    one = next(b for b in meta.split("  - .agpr_count")[1:] if "crf_decode_pipelinedILi1E" in b)
    if ".private_segment_fixed_size: 0" not in one:
        sys.exit("crf_decode_pipelined<1> uses scratch")
    # the two-tile shape (eight workgroups per CU, 64 registers)
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5gecco12_GLOBAL__N_120crf_decode_pipelinedILi2E"))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
    body = lines[start:end]
    # the tile path: from the first conditional branch (block index against the number of Viterbi blocks) to the label it skips to
    br = next(i for i, l in enumerate(body) if "s_cbranch_scc" in l)
    target = body[br].split()[-1] + ":"
    stop = next(i for i, l in enumerate(body) if l.startswith(target))
    scratch = [(i, body[i].strip()) for i in range(br, stop) if "scratch_" in body[i]]
    rcp = [i for i in range(br, stop) if "v_rcp_f64" in body[i]]
    print(f"tile path: lines {br}..{stop} of {len(body)}, {len(rcp)} v_rcp_f64, scratch accesses: {scratch}")
    if not rcp:
        sys.exit("could not locate the tile path (no v_rcp_f64 between the role branch and its target)")
    # (the spill of v0 that makes room for the Viterbi workgroups' SGPR spills: its store and its reload may both lie on
    # this side of the role branch, once each, outside the loop over the tiles)
    if len(scratch) > 2 or any("v0," not in s.replace("v0, off", "v0,") for _, s in scratch):
        sys.exit("the tile path of crf_decode_pipelined spills")


if __name__ == "__main__":
    main()
