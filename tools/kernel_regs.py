"""VGPR / SGPR / spill / LDS / code size of every kernel of one .hip source, from the gfx950 assembly's metadata.
    python tools/kernel_regs.py render.hip [filter-substring ...]        (compiles with the library's flags; no GPU needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
if not os.path.exists(src):
    src = os.path.join(ROOT, "nerfacc_amd", "csrc", src)
filt = sys.argv[2:]
extra = os.environ.get("NFA_EXTRA", "").split()
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           *extra, "-o", out, src], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    if "--keep" in os.environ.get("NFA_KEEP", ""):
        open("/tmp/kernel_regs.s", "w").write(txt)
rows = []
for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target)", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    if filt and not any(f in name for f in filt):
        continue
    rows.append((name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size")))
# code size per kernel from the symbol sizes is not in the metadata; count instructions between the label and s_endpgm instead
print("| kernel | VGPR | AGPR | SGPR | spilled | static LDS |\n|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %s | %s | %s | %s | %s |" % r)
