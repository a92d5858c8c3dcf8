#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of a built libdemc_hip.so (from the code objects' metadata notes; no GPU).
usage: tools/kernel_regs.py [lib.so] [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib, tmp):
    fat = os.path.join(tmp, "fatbin.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    out = []
    for i, a in enumerate(starts):
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = os.path.join(tmp, f"device{i}.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        for block in notes.split("- .agpr_count:")[1:]:
            g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", block).group(1))
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            out.append(dict(name=subprocess.check_output(["c++filt", name], text=True).strip(), agpr=int(block.split()[0]),
                            vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                            lds=g("group_segment_fixed_size"), wg=g("max_flat_workgroup_size"),
                            spill=g("vgpr_spill_count") if "vgpr_spill_count" in block else 0))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "differentialevolutionmcmc.jl_amd", "libdemc_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        for k in sorted(kernels(lib, tmp), key=lambda k: k["name"]):
            if args and not any(a in k["name"] for a in args):
                continue
            print(f"{k['vgpr']:4d} regs ({k['agpr']:3d} agpr) {k['sgpr']:4d} sgpr  scratch {k['scratch']:5d}  spill {k['spill']:3d}  lds {k['lds']:6d}  wg {k['wg']:4d}  {k['name'][:150]}")
