#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels of one translation unit, from hipcc's own report
(-Rpass-analysis=kernel-resource-usage).  Cross-compiles: runs in the build container, no GPU.

    python profiles/resource_usage.py env_pg.hip [filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neorl-industrial-gym_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-std=c++17"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    src = argv[0]
    flt = argv[1] if len(argv) > 1 else ""
    cmd = ["hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    recs, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?):\s+(\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            recs.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([r["name"] for r in recs])
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr':>5} {'occ':>4} {'vspill':>6} {'sspill':>6} {'LDS':>7}  kernel")
    for r, n in zip(recs, names):
        n = n.replace("nig::", "").replace("void ", "")
        n = re.sub(r"\(.*\)$", "", n)
        if flt and flt not in n:
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} {r.get('ScratchSize [bytes/lane]', '?'):>5} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4} {r.get('VGPRs Spill', '?'):>6} {r.get('SGPRs Spill', '?'):>6} "
              f"{r.get('LDS Size [bytes/block]', '?'):>7}  {n}")


if __name__ == "__main__":
    main()
