"""Register / LDS / occupancy table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: [RESUSAGE_FLAGS="-DPT_REG_MINB=3"] python tools/resusage.py cn_pointwise [cn_norm ...]   (no GPU needed)"""
import re, subprocess, sys, os, tempfile
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cultionet_amd", "csrc")
for stem in sys.argv[1:]:
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
                            "-Rpass-analysis=kernel-resource-usage", *os.environ.get("RESUSAGE_FLAGS", "").split(), "-c", os.path.join(here, stem + ".hip"), "-o", os.path.join(d, "x.o")],
                           capture_output=True, text=True)
    pat = (r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
           r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)")
    for m in re.finditer(pat, r.stderr, re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        print(f"{name[:80]:80s} vgpr={m.group(2):>3s} agpr={m.group(3):>3s} scratch={m.group(4):>4s} waves/simd={m.group(5)} lds={m.group(6)}")
