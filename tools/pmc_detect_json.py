"""tools/pmc_kernel.sh listings of the detect kernels at C2 / C3 / C5 (one '# detect kernels at Cn ...' header per configuration, then
'kernel  COUNTER  value  dur_us x' lines) -> profiles/rNN_pmc_detect_configs.json: per configuration and kernel the counters, the mean duration and
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (the x2 = the gfx950 wide-read correction of MI355X_MICROARCH.md).  bench.py reads the newest of
these files for `configs.C2 / C3 / C5.frac` (whole detect pass by counter bytes).
   python tools/pmc_detect_json.py gpurun_out/prof/r06_pmc_detect_configs.txt > profiles/r06_pmc_detect_configs.json"""
import json, re, sys
FRAMES = {"C2": 256, "C3": 256, "C5": 64}
out, cur = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"# detect kernels at (C\d)", line)
    if m:
        cur = out.setdefault(m.group(1), {"frames_per_launch": FRAMES[m.group(1)], "kernels": {}})
        continue
    m = re.match(r"(\S.*?)\s{2,}(\w+)\s+([0-9.e+]+) dur_us ([0-9.]+)", line)
    if m and cur is not None:
        k = cur["kernels"].setdefault(m.group(1).strip(), {"_dur": []})
        k[m.group(2)] = float(m.group(3)); k["_dur"].append(float(m.group(4)))
for cfg, d in out.items():
    tot = t = 0.0
    for name, k in d["kernels"].items():
        k["dur_us"] = sum(k["_dur"]) / len(k["_dur"]); del k["_dur"]
        k["hbm_bytes"] = (2 * k.get("FETCH_SIZE", 0.0) + k.get("WRITE_SIZE", 0.0)) * 1024
        tot += k["hbm_bytes"]; t += k["dur_us"]
    d["pass_hbm_bytes"] = tot; d["kernel_time_us"] = t
    d["note"] = ("HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; the x2 is the gfx950 wide-read correction of MI355X_MICROARCH.md), separate --pmc passes of "
                 "bench.py --config C --stages segment --force-all --no-pipeline")
print(json.dumps(out, indent=1))
