#!/usr/bin/env python3
"""make_traffic_json.py PREFIX OUT.json: per-kernel HBM-side traffic per launch from the FETCH_SIZE / WRITE_SIZE passes of
tools/prof_r03.sh (PREFIX_pmc_fetch_g1.txt, ..._write_g1.txt, ..._fetch_g8.txt, ..._write_g8.txt: per-kernel means in KB).
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE is taken as it is."""
import json
import re
import sys

prefix, out = sys.argv[1], sys.argv[2]
KERNELS = ("k_enc_p", "k_bproj_p", "k_scan_pairl_asm", "k_cgate_p", "k_resid_minmax16", "k_dec_p")


def read(path, ctr):
    vals = {}
    for line in open(path):
        m = re.search(ctr + r"=([0-9.e+]+)", line)
        if not m:
            continue
        for k in KERNELS:
            if k + "<" in line or k + "(" in line:
                vals[k] = float(m.group(1))
    return vals


try:  # the bench line of the same profiling run says how many state slots the recurrence streams of each layer held
    line = [l for l in open(f"{prefix}_g8_single.json") if l.startswith("{")][-1]
    stream_slots = json.loads(line)["roofline"]["stream_slots_per_layer"]
except (OSError, KeyError, IndexError):
    stream_slots = [32, 32, 32]
doc = dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over bench.py --inflight 1 (tools/prof_r03.sh); "
                "bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB x 1000); B=32, L=4096, dim_scale 0.5, layers compacted to 32 state slots, recurrence streams on the live ones (stream_slots)",
           B=32, L=4096, P=64, stream_slots=stream_slots, entries=[])
for g in (1, 8):
    f, w = read(f"{prefix}_pmc_fetch_g{g}.txt", "FETCH_SIZE"), read(f"{prefix}_pmc_write_g{g}.txt", "WRITE_SIZE")
    per = {k: int(round((2 * f[k] + w[k]) * 1000)) for k in KERNELS if k in f and k in w}
    # three layers; the last one's residual pass is part of the decoder (proj_p.hpp k_dec_p<.., RESID>): two launches of k_resid_minmax16
    fwd = per["k_enc_p"] + 3 * (per["k_bproj_p"] + per["k_scan_pairl_asm"] + per["k_cgate_p"]) + 2 * per["k_resid_minmax16"] + per["k_dec_p"]
    doc["entries"].append(dict(batches_per_launch=g, traffic_bytes_per_launch=per, per_launch_set=fwd, per_batch=fwd // g,
                               FETCH_SIZE_KB=f, WRITE_SIZE_KB=w))
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({e["batches_per_launch"]: e["per_batch"] for e in doc["entries"]}))
