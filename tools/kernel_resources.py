"""Register / scratch / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py temporalstereo_amd/csrc/block_cost.hip [name filter]
"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-munsafe-fp-atomics", "-fno-slp-vectorize"]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-90s %5s %5s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "occ"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt in n:
            print("%-90s %5d %5d %7d %4d" % (n[:90], r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("occ", -1)))


if __name__ == "__main__":
    main()
