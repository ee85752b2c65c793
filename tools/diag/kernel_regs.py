"""(CPU) VGPR / SGPR / LDS / scratch of every kernel of the built library: python tools/diag/kernel_regs.py [lib] [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "fateavatar_amd", "libfr_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
    for k, st in enumerate(starts):
        en = starts[k + 1] if k + 1 < len(starts) else len(data)
        b, co = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"d{k}.co")
        open(b, "wb").write(data[st:en])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={b}", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        cur = {}
        for line in notes.splitlines():
            line = line.strip()
            for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".agpr_count:", ".group_segment_fixed_size:", ".private_segment_fixed_size:"):
                if line.startswith(key):
                    cur[key[1:-1]] = line.split()[-1]
            if line.startswith(".wavefront_size:") and cur.get("name", "").startswith("_ZN2fr") and flt in cur["name"]:
                print(f"{cur['name'][:70]:70s} vgpr {cur.get('vgpr_count')} agpr {cur.get('agpr_count')} sgpr {cur.get('sgpr_count')} lds {cur.get('group_segment_fixed_size')} scratch {cur.get('private_segment_fixed_size')}")
