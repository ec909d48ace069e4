"""Static guard for DESIGN.md section 2b (no GPU needed): the shipped code objects are disassembled and checked for the
packed-fp32 instruction forms that were measured to return wrong values in lanes 48..63 beside another kernel's dense
f16 MFMAs (tools/pk_probe.py: v_pk_add_f32 with an op_sel half-swizzle on a vector-register source; plain, negated,
scalar-operand and constant-operand forms were not affected in 1e11 checked operations each).

1. Producer kernels -- anything the pipeline or a caller may run on a stream of its own beside the sweeps -- contain
   no packed-fp32 instruction at all (csrc/Makefile NOPK_OBJS).
2. No kernel of the library contains a packed-fp32 instruction with an `op_sel:` modifier; the sweeps keep their plain
   and negated forms."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "audio_tokens_amd" / "libaudio_tokens_amd.so"
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
PRODUCERS = ("logmel_kernel", "logmel_any_kernel", "resample_kernel", "resample_tiled_kernel", "l2norm_rows_kernel",
             "minmax_scale_kernel", "minmax_apply_kernel", "minmax_init_kernel", "conv1d_mel_kernel", "pairwise_sumsq")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not LIB.exists() or not OBJDUMP.exists():
        pytest.skip("needs the built library and ROCm's llvm-objdump")
    work = tmp_path_factory.mktemp("codeobj")
    shutil.copy(LIB, work / LIB.name)
    subprocess.run([str(OBJDUMP), "--offloading", LIB.name], cwd=work, check=True, capture_output=True)
    objs = sorted(work.glob("*gfx950*"))
    assert objs, "no gfx950 code object found in the library"
    out = {}
    for o in objs:
        dis = subprocess.run([str(OBJDUMP), "-d", str(o)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                out.setdefault(name, [])
            elif name is not None and "v_pk_" in line:
                out[name].append(line.split("//")[0].strip())
    assert any("assign_f16filter_kernel" in n for n in out), "the sweeps were not found: wrong library?"
    return out


def test_producers_hold_no_packed_fp32_instruction(kernels):
    found = {p: [n for n in kernels if p in n] for p in PRODUCERS}
    for p in ("logmel_kernel", "logmel_any_kernel", "resample", "l2norm_rows_kernel", "conv1d_mel_kernel"):
        assert any(p in n for n in kernels), f"{p}: not in the library"
    for p, names in found.items():
        for n in names:
            assert not kernels[n], f"{n}: {len(kernels[n])} packed-fp32 instructions, e.g. {kernels[n][0]}"


def test_no_packed_fp32_instruction_with_a_half_swizzle_anywhere(kernels):
    total = 0
    for n, ins in kernels.items():
        total += len(ins)
        for i in ins:
            assert "op_sel:" not in i, f"{n}: {i}"
    assert total > 1000          # (the sweeps do use packed ops: the disassembly was read)
