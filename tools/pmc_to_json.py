#!/usr/bin/env python3
"""rocprofv3 PMC passes (rocpd sqlite, one pass per counter group: tools/profile.sh) -> profiles/pmc_traffic.json,
the per-launch counter figures bench.py quotes next to its live timings.
usage: pmc_to_json.py <dir-with-pmc*/ dbs> <workload[:pipelined]> <tag> [kernel-substring]
(`C3` = the window kernel from passes of `bench.py --windowed-only`; `C3:pipelined` = crf_decode_pipelined from passes of
the default schedule)"""
import datetime
import glob
import json
import os
import re
import sqlite3
import sys


def main():
    root, workload, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    pat = sys.argv[4] if len(sys.argv) > 4 else "crf_windowed_l2"
    vals, kernel = {}, None
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        try:
            rows = list(con.execute(
                "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                "where kernel_name like ? group by kernel_name, counter_name", (f"%{pat}%",)))
        except sqlite3.Error:
            rows = []
        for k, c, v, n in rows:
            vals[c] = float(v)
            mm = re.search(r"(crf_\w+<[^>]*>|crf_\w+|\w+)\(", k.replace("(anonymous namespace)::", ""))
            kernel = mm.group(1) if mm else k
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE for a kernel matching {pat!r} under {root}")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_path = os.path.join(repo, "profiles", "pmc_traffic.json")
    import hashlib

    sha = hashlib.sha256()
    for name in ("crf_kernels.hip", "crf_device.hpp", "crf_vd_short.hpp", "crf_scan.hpp"):
        with open(os.path.join(repo, "gecco_amd", "csrc", name), "rb") as fh:
            sha.update(fh.read())
    try:
        doc = json.load(open(out_path))
    except Exception:
        doc = {}
    entry = {
        # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE are in KiB;
        # gfx950 under-reports the read side by 2x (64-B requests counted as 32 B)
        "hbm_bytes_per_launch": int(round(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)),
        "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
        "kernel": kernel,
        "source": f"profiles/{tag}_pmc.json <- {os.path.relpath(os.path.abspath(root), repo)}/pmc*: rocprofv3 --pmc passes of "
                  f"`bench.py{'' if ':' in workload else ' --windowed-only'}` ({datetime.date.today().isoformat()}), FETCH_SIZE x2 (gfx950 read-side correction) "
                  f"+ WRITE_SIZE",
        "kernel_source_sha16": sha.hexdigest()[:16],  # bench.py drops these figures when the kernel source has changed since
    }
    for c in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT",
              "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if c in vals:
            entry[c] = vals[c]
    doc[workload] = entry
    json.dump(doc, open(out_path, "w"), indent=1)
    tag_path = os.path.join(os.path.dirname(out_path), f"{tag}_pmc.json")
    try:
        tag_doc = json.load(open(tag_path))
    except Exception:
        tag_doc = {}
    tag_doc[workload] = dict(entry, all_counters=vals)
    json.dump(tag_doc, open(tag_path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
