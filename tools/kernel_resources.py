#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy report of one HIP source (dev tool):
    python tools/kernel_resources.py viet-asr_amd/csrc/encoder_pw_split.hip [substring ...]
Compiles for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line per kernel instantiation."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, filt = sys.argv[1], sys.argv[2:]
extra = [a for a in filt if a.startswith("-")]
filt = [a for a in filt if not a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{ROOT}/viet-asr_amd/csrc",
       "-ffp-contract=fast", "-c", src, "-o", "/tmp/kr.o", "-Rpass-analysis=kernel-resource-usage"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
names = {}
for b in re.split(r"(?=remark: [^\n]*Function Name)", txt):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    names[m.group(1)] = (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                         g(r"LDS Size \[bytes/block\]"), g("SGPRs"))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
print(f"{'kernel':90s} vgpr agpr scratch occ lds sgpr")
for d, (n, r) in zip(dem, names.items()):
    d = d.replace("void vasr::(anonymous namespace)::", "").replace("vasr::PwArgs, int, int, int", "...")
    if not filt or any(f in d for f in filt):
        print(f"{d[:90]:90s} " + " ".join(r))
