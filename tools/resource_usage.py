"""Per-kernel register / scratch / occupancy table of a HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py chainpartitioners.jl_amd/csrc/dp_total.hip [--all]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except FileNotFoundError:
        return names


def main():
    src = sys.argv[1]
    show_all = "--all" in sys.argv
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    rows = []
    for b in blocks:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append([name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")])
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'vspill':>7s} {'sspill':>7s} {'occ':>4s} {'LDS':>6s}")
    for r, nm in zip(rows, names):
        if show_all or r[3] > 0 or r[4] > 0 or r[5] > 0:
            nm = re.sub(r"\(.*", "", nm)[:100]
            print(f"{nm:100s} {r[1]:5d} {r[2]:5d} {r[3]:8d} {r[4]:7d} {r[5]:7d} {r[6]:4d} {r[7]:6d}")


if __name__ == "__main__":
    main()
