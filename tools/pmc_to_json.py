#!/usr/bin/env python3
"""pmc_traffic.txt (tools/pmc_summary.py over a --pmc FETCH_SIZE pass and a --pmc WRITE_SIZE pass of `bench.py --steps N
--warmup 0`) -> profiles/pmc_traffic.json: HBM bytes per launch of every library kernel and per bench step.
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: the counters are in KiB;
FETCH_SIZE tallies 128-byte requests as 64 bytes, so it is doubled; WRITE_SIZE is exact.
usage: pmc_to_json.py pmc_traffic.txt STEPS SOURCE_NAME > pmc_traffic.json"""
import json
import re
import sys


def main():
    path, steps, source = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+launches (\d+)$", line.rstrip())
        if m:
            name = m.group(1).replace("void ", "").split("(")[0].strip()
            cur = kernels.setdefault(name, {"launches": int(m.group(2)), "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] += float(m.group(2))
    sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
    import bench
    out = {"csrc_sha256": bench.csrc_sha256(), "workload": "bench.py --steps %d --warmup 0 (8,000,000 reads/step, qb 29: the whole build with its deNoise points)" % steps,
           "source": source, "units": "bytes; FETCH_SIZE (KiB) x 1024 x 2 (gfx950 correction), WRITE_SIZE (KiB) x 1024", "kernels": {}}
    total = 0.0
    for name, v in kernels.items():
        if not name.startswith("k_"):
            continue        # torch kernels that generate the synthetic text are not part of the path
        f, w = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
        out["kernels"][name] = {"launches": v["launches"], "fetch_bytes_per_launch": int(f / v["launches"]),
                                "write_bytes_per_launch": int(w / v["launches"])}
        total += f + w
    out["bytes_per_step"] = int(total / steps)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
