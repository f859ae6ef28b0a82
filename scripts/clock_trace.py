#!/usr/bin/env python3
"""Clock / power trace of the GPU while a command runs (VERDICT r05 evidence gap 8a: what do the clocks do under the sampling loop and
under a matrix-pipe-only loop?).  Samples the amdgpu hwmon files of the first GPU every 50 ms -- freq1_input (shader clock, Hz),
power1_average / power1_input (uW), temp -- with `rocm-smi --showclocks --showpower --json` as the fallback when sysfs is not
readable, for 2 s of idle, then for the lifetime of the child command.

    python scripts/clock_trace.py out.txt -- python bench.py --steps 100 --warmup 5 --no-profile --no-cpu-baseline --no-secondary
"""
import glob
import json
import os
import statistics
import subprocess
import sys
import time


def hwmon_files():
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f = {}
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                f[name] = p
        if "freq1_input" in f:
            return f
    return {}


def read_sysfs(files):
    out = {}
    for k, p in files.items():
        try:
            out[k] = int(open(p).read().strip())
        except Exception:
            pass
    return out


def read_smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        out = {}
        for k, v in card.items():
            lk = k.lower()
            if "sclk" in lk and "mhz" in str(v).lower():
                out["freq1_input"] = int(float(str(v).lower().replace("(", "").replace(")", "").replace("mhz", "").strip()) * 1e6)
            if "power" in lk and "w" in lk:
                try:
                    out["power1_average"] = int(float(v) * 1e6)
                except Exception:
                    pass
        return out
    except Exception:
        return {}


def main():
    out_path, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    files = hwmon_files()
    read = (lambda: read_sysfs(files)) if files else read_smi
    period = 0.05 if files else 0.5
    rows = []

    def sample(tag):
        s = read()
        s["t"], s["phase"] = time.time(), tag
        rows.append(s)

    t_end = time.time() + 2.0
    while time.time() < t_end:
        sample("idle")
        time.sleep(period)
    child = subprocess.Popen(cmd)
    while child.poll() is None:
        sample("run")
        time.sleep(period)
    with open(out_path, "w") as f:
        f.write(f"# {' '.join(cmd)}\n# source: {'sysfs hwmon ' + str(files) if files else 'rocm-smi --json'}; period {period} s; exit code {child.returncode}\n")
        for ph in ("idle", "run"):
            for key, scale, unit in (("freq1_input", 1e6, "MHz shader clock"), ("power1_average", 1e6, "W"), ("power1_input", 1e6, "W (input)"),
                                     ("temp1_input", 1e3, "C")):
                v = [r[key] / scale for r in rows if r["phase"] == ph and key in r]
                if v:
                    q = statistics.quantiles(v, n=10) if len(v) >= 10 else [min(v)] * 9
                    f.write(f"{ph:5s} {unit:18s} n={len(v):5d} min {min(v):8.1f} p10 {q[0]:8.1f} median {statistics.median(v):8.1f} p90 {q[8]:8.1f} max {max(v):8.1f}\n")
        f.write("# t_rel_s phase MHz W\n")
        t0 = rows[0]["t"]
        for r in rows:
            f.write(f"{r['t'] - t0:8.2f} {r['phase']} {r.get('freq1_input', 0) / 1e6:8.1f} {r.get('power1_average', r.get('power1_input', 0)) / 1e6:8.1f}\n")
    print(open(out_path).read().split("# t_rel_s")[0])
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
