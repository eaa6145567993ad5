"""Register / LDS / scratch use of every kernel in a hipcc object or shared library (reads the embedded gfx950 code objects'
metadata notes with llvm-readelf): python tools/kernel_resources.py crdr_amd/_lib/igemm_p0.o [...]"""
import re
import struct
import subprocess
import sys
import tempfile


def code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            return
        if struct.unpack_from("<H", data, i + 18)[0] == 224:   # EM_AMDGPU
            shoff = struct.unpack_from("<Q", data, i + 40)[0]
            shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
            yield data[i:i + shoff + shentsize * shnum]
        pos = i + 4


def kernels(path):
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for b in out.split("- .agpr_count")[1:]:
            g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1)) if re.search(r"\." + k + r":\s+(\d+)", b) else -1
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            yield {"name": name, "agpr": int(re.match(r":\s+(\d+)", b).group(1)), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                   "spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")}


if __name__ == "__main__":
    for p in sys.argv[1:]:
        for k in kernels(p):
            dem = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            print(f"vgpr {k['vgpr']:4d} (agpr {k['agpr']:3d}) sgpr {k['sgpr']:4d} spill {k['spill']:3d} scratch {k['scratch']:5d} lds {k['lds']:6d}  {dem}")
