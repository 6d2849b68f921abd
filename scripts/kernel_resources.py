"""Prints VGPR/SGPR/scratch/occupancy per kernel from hipcc's
-Rpass-analysis=kernel-resource-usage (developer tool)."""
import re, subprocess, sys, os
SRC = "kernels.hip"
if len(sys.argv) > 1 and sys.argv[1].endswith(".hip"):
    SRC = sys.argv.pop(1)
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "thz_image_explorer_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
                      SRC, "-o", "/tmp/_k.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                     cwd=src, stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("thz::", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':60s} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'spillV':>7} {'occ':>4}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('TotalSGPRs',0):5d} "
          f"{r.get('ScratchSize',0):8d} {r.get('VGPRs Spill',0):7d} {r.get('Occupancy',0):4d}")
