"""Register / scratch / LDS usage of the shipped kernels, read from the code objects inside ``csrc/liblrt_hip.so``.

Why this exists: ``k_fwd_cr4`` reads registers across lanes (``v_readlane``, DPP, ``ds_bpermute``) in
Phase B.  A VGPR that the register allocator spills to scratch is stored and reloaded under the EXEC mask of the place where the
spill code lands; inside a divergent region the inactive lanes keep stale contents, and a later cross-lane read returns them
(a dense scene faulted that way in round 3, and the 16-wave instantiation of round 4 faulted for the same reason).  Nothing
enforced "no spills" -- a compiler update could turn the kernel into silent garbage.  ``check()`` does: it is run by
``lidar_rt_amd.build.build()`` after every compile (and by ``__graft_entry__.build()``), and fails the build when an
instantiation of a gated kernel reports spilled VGPRs or private (scratch) memory.  SGPR spills are not an error: they go to
VGPR lanes with ``v_writelane`` / ``v_readlane`` (which ignore EXEC), never to memory.

The numbers come from the ``NT_AMDGPU_METADATA`` note (msgpack) of each gfx950 code object of the clang offload bundles in the
library's ``.hip_fatbin`` section -- no external tool is needed.
"""
from __future__ import annotations

import re
import struct
import subprocess
from typing import Dict, List

# kernels that must not spill a vector register or use scratch: regular expression on the demangled name
GATED = (r"^k_fwd_cr4<",)
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path: str) -> List[bytes]:
    """The device ELFs (triple hip*-amdgcn-...) of every offload bundle in the shared library."""
    blob = open(lib_path, "rb").read()
    out = []
    for m in re.finditer(re.escape(_BUNDLE_MAGIC), blob):
        p0 = m.start()
        (n,) = struct.unpack_from("<Q", blob, p0 + 24)
        off = p0 + 32
        if n > 64:
            continue
        for _ in range(n):
            o, sz, ts = struct.unpack_from("<QQQ", blob, off); off += 24
            triple = blob[off:off + ts].decode("ascii", "replace"); off += ts
            if triple.startswith("hip") and "amdgcn" in triple and sz > 0:
                out.append(blob[p0 + o:p0 + o + sz])
    return out


def _notes(elf: bytes):
    """(name, type, desc) of every note of an ELF64 little-endian object."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        raise ValueError("not an ELF64 object")
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        b = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, b + 4)
        if sh_type != 7:                                              # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, b + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p); p += 12
            name = elf[p:p + namesz].rstrip(b"\0").decode("ascii", "replace"); p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]; p += (descsz + 3) & ~3
            yield name, ntype, desc


def _demangle(names: List[str]) -> List[str]:
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60)
        out = r.stdout.split("\n")[:len(names)]
        if r.returncode == 0 and len(out) == len(names):
            return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", o)) for o in out]
    except (OSError, subprocess.SubprocessError):
        pass
    return names


def kernel_resources(lib_path: str) -> Dict[str, dict]:
    """{demangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_bytes, max_workgroup, waves_per_simd,
    workgroups_per_cu}} for every kernel of the library.  Occupancy: the 512-entry unified register file of a gfx950 SIMD in granules of
    8, at most 8 waves per SIMD; 160 KB of LDS per CU."""
    import msgpack
    rows, mangled = [], []
    for elf in code_objects(lib_path):
        for name, ntype, desc in _notes(elf):
            if name != "AMDGPU" or ntype != 32:                        # NT_AMDGPU_METADATA
                continue
            md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
            for k in md.get("amdhsa.kernels", []):
                mangled.append(k[".name"]); rows.append(k)
    res = {}
    for nm, k in zip(_demangle(mangled), rows):
        v, a = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
        alloc = max(8, (v + a + 7) // 8 * 8)
        wg = int(k.get(".max_flat_workgroup_size", 0)); lds = int(k.get(".group_segment_fixed_size", 0))
        waves = min(8, 512 // alloc)
        waves_wg = max(1, (wg + 63) // 64)
        by_regs = (4 * waves) // waves_wg
        by_lds = (160 * 1024) // lds if lds > 0 else 1 << 30
        res[nm] = {"vgpr": v, "agpr": a, "sgpr": int(k.get(".sgpr_count", 0)), "vgpr_spill": int(k.get(".vgpr_spill_count", 0)),
                   "sgpr_spill": int(k.get(".sgpr_spill_count", 0)), "scratch_bytes": int(k.get(".private_segment_fixed_size", 0)),
                   "lds_bytes": lds, "max_workgroup": wg, "waves_per_simd": waves,
                   "workgroups_per_cu": int(min(by_regs, by_lds, 32)), "dynamic_stack": bool(k.get(".uses_dynamic_stack", False))}
    return res


def violations(res: Dict[str, dict], gated=GATED) -> List[str]:
    bad = []
    for name, r in sorted(res.items()):
        if any(re.search(g, name) for g in gated) and (r["vgpr_spill"] or r["scratch_bytes"] or r["dynamic_stack"]):
            bad.append(f"{name}: {r['vgpr_spill']} spilled VGPRs, {r['scratch_bytes']} B of scratch per lane"
                       + (", dynamic stack" if r["dynamic_stack"] else ""))
    return bad


def check(lib_path: str, gated=GATED) -> Dict[str, dict]:
    """Raise RuntimeError when a gated kernel spills vector registers or uses scratch; returns the table otherwise."""
    res = kernel_resources(lib_path)
    if not any(re.search(g, n) for g in gated for n in res):
        raise RuntimeError(f"{lib_path}: no kernel matches {gated}: the resource gate checked nothing")
    bad = violations(res, gated)
    if bad:
        raise RuntimeError("kernel resource gate: these instantiations must not spill vector registers (cross-lane reads of a "
                           "register reloaded under a partial EXEC mask return stale lanes):\n  " + "\n  ".join(bad))
    return res


def table_md(res: Dict[str, dict], pattern: str = r".") -> str:
    rows = ["| kernel | VGPR | AGPR | SGPR | spilled VGPR | spilled SGPR (to VGPR lanes) | scratch B | LDS B | workgroup | waves / SIMD | workgroups / CU |",
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for n, r in sorted(res.items()):
        if re.search(pattern, n):
            rows.append(f"| `{n}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['vgpr_spill']} | {r['sgpr_spill']} | {r['scratch_bytes']} | "
                        f"{r['lds_bytes']} | {r['max_workgroup']} | {r['waves_per_simd']} | {r['workgroups_per_cu']} |")
    return "\n".join(rows)


if __name__ == "__main__":
    import os
    import sys
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblrt_hip.so")
    table = check(lib)
    print(table_md(table, sys.argv[2] if len(sys.argv) > 2 else r"."))
