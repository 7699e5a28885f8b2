"""Static checks on the SHIPPED gfx950 code of the pooled CG kernel.

solver_cgp.hpp issues its matrix-vector products as `v_fmac_f32_dpp ... row_newbcast` from inline assembly; the compiler's hazard
recogniser does not look inside inline asm, so the DPP data hazards of the CDNA ISA are the source's responsibility (`dpp_fence`: an
`s_nop 4` tied to the broadcast registers).  What register allocation and scheduling made of it is checked here on the binary:

  * a DPP instruction's src0 VGPR must not be written by a VALU instruction within the 2 preceding wait states;
  * EXEC must not be written by a VALU instruction (v_cmpx, v_readlane-class excluded: SALU writes need 0) within the 5 preceding wait states.
An instruction is one wait state, `s_nop N` is N + 1.  The walk follows straight-line order and stops at labels (a branch target's
predecessors are not all visible: the source never branches between the fence and the products)."""

import glob
import os
import re
import struct
import subprocess
import tempfile

import pytest

import conftest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _gfx950_disassembly(path):
  b = open(path, "rb").read()
  out = []
  for mm in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", b):
    i = mm.start()
    p = i + 24
    nb = struct.unpack("<Q", b[p:p + 8])[0]
    p += 8
    for _ in range(nb):
      off, size, tl = struct.unpack("<QQQ", b[p:p + 24])
      p += 24
      tr = b[p:p + tl].decode()
      p += tl
      if "gfx950" not in tr:
        continue
      with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(b[i + off:i + off + size])
      out.append(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout)
      os.unlink(f.name)
  return "\n".join(out)


def _vgprs(tok):
  """VGPR numbers named by one operand token: v12 or v[12:15]."""
  mm = re.fullmatch(r"v(\d+)", tok)
  if mm:
    return {int(mm.group(1))}
  mm = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
  return set(range(int(mm.group(1)), int(mm.group(2)) + 1)) if mm else set()


def _check_kernel(lines):
  """lines: the instructions of one kernel in order ('' marks a label).  Returns (number of DPP products, list of violations)."""
  ndpp, bad = 0, []
  for i, ln in enumerate(lines):
    if not ln.startswith("v_fmac_f32_dpp"):
      continue
    ndpp += 1
    ops = [t.strip(",") for t in ln.split()[1:4]]
    src0 = _vgprs(ops[1])
    states, j = 0, i - 1
    while j >= 0 and states < 5 and lines[j] != "":
      prev = lines[j]
      op = prev.split()[0]
      if op.startswith("v_") and not op.startswith("v_fmac_f32_dpp"):
        dst = _vgprs(prev.split()[1].strip(",")) if len(prev.split()) > 1 else set()
        if states < 2 and dst & src0:
          bad.append((i, ln, prev, states))
        if op.startswith("v_cmpx") and states < 5:
          bad.append((i, ln, prev, states))
      elif op.startswith("v_fmac_f32_dpp"):
        if _vgprs(prev.split()[1].strip(",")) & src0 and states < 2:
          bad.append((i, ln, prev, states))
      mm = re.match(r"s_nop\s+(\d+)", prev)
      states += int(mm.group(1)) + 1 if mm else 1
      j -= 1
  return ndpp, bad


def test_inline_asm_dpp_products_keep_their_wait_states():
  objs = sorted(glob.glob(os.path.join(conftest.ROOT, "build", "objcache", "solve_cgp.hip.*.o")))
  path = objs[-1] if objs else os.path.join(conftest.ROOT, "mujoco_warp_amd", "libmjhip.so")
  if not os.path.exists(path) or not os.path.exists(OBJDUMP):
    pytest.skip("no built object / no llvm-objdump")
  text = _gfx950_disassembly(path)
  kernels = {}
  for blk in re.split(r"\n(?=[0-9a-f]+ <)", text):
    head = blk.split("\n", 1)[0]
    if "k_solve_cgp_plus" not in head:
      continue
    name = re.search(r"<(\S+)>", head).group(1)
    lines = []
    for l in blk.split("\n")[1:]:
      t = l.strip()
      if not t or t.startswith(";"):
        continue
      if t.startswith("<") or t.endswith(":"):
        lines.append("")  # a label
        continue
      lines.append(t.split("//")[0].strip())
    kernels[name] = lines
  assert len(kernels) == 8, sorted(kernels)  # NV4 = 1 .. 8
  for name, lines in kernels.items():
    ndpp, bad = _check_kernel(lines)
    assert ndpp >= 50, (name, ndpp)  # (the check found the products it is about)
    assert not bad, (name, bad[:3])


def test_the_checker_sees_a_planted_hazard():
  ok = ["s_nop 4", "v_fmac_f32_dpp v8, v6, v60 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
  assert _check_kernel(ok) == (1, [])
  planted = ["v_mov_b32_e32 v6, v3", "s_nop 0", "v_fmac_f32_dpp v8, v6, v60 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
  assert len(_check_kernel(planted)[1]) == 1
  far = ["v_mov_b32_e32 v6, v3", "s_nop 1", "v_fmac_f32_dpp v8, v6, v60 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
  assert _check_kernel(far)[1] == []
  pair = ["v_pk_mul_f32 v[6:7], v[2:3], v[4:5]", "v_fmac_f32_dpp v8, v6, v60 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1"]
  assert len(_check_kernel(pair)[1]) == 1
