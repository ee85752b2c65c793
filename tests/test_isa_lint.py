"""What the compiler made of the kernels in the BUILT library (fateavatar_amd/libfr_hip.so), checked on the CPU: the gfx950
code objects are taken out of the shared object and disassembled.  Two findings of round 3 are what this guards: LDS words
reached through `volatile` or through a pointer that may be LDS or global memory compile to FLAT loads and stores (slower,
and their s_waitcnt vmcnt(0) also waits for every global access in flight), and small arrays indexed by a loop variable the
compiler does not unroll, or conditional assignments between HIP float4 structs, go through scratch memory."""
import os
import re
import shutil
import subprocess

import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fateavatar_amd", "libfr_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LLVM, t) for t in ("clang-offload-bundler", "llvm-objdump", "llvm-readelf")]


def _code_objects(tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    out = []
    for k, st in enumerate(starts):
        en = starts[k + 1] if k + 1 < len(starts) else len(data)
        b, co = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"d{k}.co")
        open(b, "wb").write(data[st:en])
        subprocess.run([TOOLS[0], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={b}",
                        f"--output={co}"], check=True)
        out.append(co)
    return out


@pytest.mark.skipif(not os.path.exists(LIB) or not all(os.path.exists(t) for t in TOOLS) or shutil.which("objcopy") is None,
                    reason="needs the built library and the ROCm LLVM tools")
def test_no_kernel_uses_scratch_or_flat_accesses(tmp_path):
    kernels = {}
    for co in _code_objects(str(tmp_path)):
        notes = subprocess.run([TOOLS[2], "--notes", co], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split()[-1]
                kernels.setdefault(name, {})
            for key in (".private_segment_fixed_size:", ".vgpr_count:", ".uses_dynamic_stack:"):
                if line.startswith(key) and name:
                    kernels[name][key[1:-1]] = line.split()[-1]
        dis = subprocess.run([TOOLS[1], "-d", co], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                continue
            if cur in kernels and re.search(r"\b(flat_(load|store|atomic)\w*|scratch_(load|store)\w*)", line):
                kernels[cur].setdefault("bad", []).append(line.strip()[:80])
    ours = {k: v for k, v in kernels.items() if k.startswith("_ZN2fr")}
    assert len(ours) >= 25, sorted(ours)                     # every kernel of the library was found
    for k, v in sorted(ours.items()):
        assert v.get("private_segment_fixed_size") == "0" and v.get("uses_dynamic_stack") == "false", (k, v)
        assert "bad" not in v, (k, v["bad"][:4])
    # register budget the occupancy of the per-Gaussian forward kernel depends on: three waves per SIMD, which is what its
    # LDS (the wave's 12 KB SH block) allows — the thread's SH row is held in 48 registers across the counting pass
    for k, v in ours.items():
        if "k_preprocess_fwd" in k:
            assert int(v["vgpr_count"]) <= 168, (k, v["vgpr_count"])
