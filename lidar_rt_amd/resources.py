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

# kernels that must not spill a vector register or use scratch: regular expressions on the (demangled) name.  Round 6: EVERY kernel of this
# project in the shipped library (k_* tracer, kc_* Chamfer / k-NN, k_pp_* pre-processing) -- rocPRIM's own kernels are not ours to gate
GATED = (r"^k_", r"^kc_")
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def is_own_kernel(name: str) -> bool:
    """A kernel defined by this project (not one of rocPRIM's instantiations); works on demangled and on mangled names."""
    return not (name.startswith("rocprim::") or name.startswith("_ZN7rocprim") or name.startswith("void rocprim::"))


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


def _demangle_plain(name: str) -> str:
    """Without c++filt: the base name of an Itanium-mangled free function -- `_Z9k_fwd_cr4ILb1ELi4ELb0EEvT_...` -> `k_fwd_cr4<...>` (template
    arguments are not decoded; "<" marks an instantiation so that the gate's `^k_fwd_cr4<` style patterns still match)."""
    m = re.match(r"^_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1)); a = m.end()
    base = name[a:a + n]
    return base + ("<" + name[a + n + 1:] + ">" if name[a + n:a + n + 1] == "I" else "")


def _demangle(names: List[str]) -> List[str]:
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60)
        out = r.stdout.split("\n")[:len(names)]
        if r.returncode == 0 and len(out) == len(names):
            return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", o)) for o in out]
    except (OSError, subprocess.SubprocessError):
        pass
    return [_demangle_plain(n) for n in names]          # c++filt is not installed: base names are enough for the gate


def _unpack_msgpack(buf: bytes):
    """The code-object metadata note is msgpack.  The `msgpack` module does the decoding where it is installed; this is the subset the note uses
    (maps, arrays, strings, integers, booleans, nil, floats, bin), so that the build's gate needs no undeclared dependency."""
    try:
        import msgpack
        return msgpack.unpackb(buf, raw=False, strict_map_key=False)
    except ImportError:
        pass
    pos = 0

    def rd(fmt, n):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)[0]; pos += n
        return v

    def take(n):
        nonlocal pos
        b = buf[pos:pos + n]; pos += n
        return b

    def obj():
        t = rd("B", 1)
        if t <= 0x7f: return t
        if t >= 0xe0: return t - 256
        if 0x80 <= t <= 0x8f: return {obj(): obj() for _ in range(t & 15)}
        if 0x90 <= t <= 0x9f: return [obj() for _ in range(t & 15)]
        if 0xa0 <= t <= 0xbf: return take(t & 31).decode("utf-8", "replace")
        if t == 0xc0: return None
        if t == 0xc2: return False
        if t == 0xc3: return True
        if t in (0xc4, 0xc5, 0xc6): return bytes(take(rd({0xc4: "B", 0xc5: ">H", 0xc6: ">I"}[t], {0xc4: 1, 0xc5: 2, 0xc6: 4}[t])))
        if t == 0xca: return rd(">f", 4)
        if t == 0xcb: return rd(">d", 8)
        if t in (0xcc, 0xcd, 0xce, 0xcf): return rd({0xcc: "B", 0xcd: ">H", 0xce: ">I", 0xcf: ">Q"}[t], {0xcc: 1, 0xcd: 2, 0xce: 4, 0xcf: 8}[t])
        if t in (0xd0, 0xd1, 0xd2, 0xd3): return rd({0xd0: "b", 0xd1: ">h", 0xd2: ">i", 0xd3: ">q"}[t], {0xd0: 1, 0xd1: 2, 0xd2: 4, 0xd3: 8}[t])
        if t in (0xd9, 0xda, 0xdb): return take(rd({0xd9: "B", 0xda: ">H", 0xdb: ">I"}[t], {0xd9: 1, 0xda: 2, 0xdb: 4}[t])).decode("utf-8", "replace")
        if t in (0xdc, 0xdd): return [obj() for _ in range(rd(">H" if t == 0xdc else ">I", 2 if t == 0xdc else 4))]
        if t in (0xde, 0xdf): return {obj(): obj() for _ in range(rd(">H" if t == 0xde else ">I", 2 if t == 0xde else 4))}
        raise ValueError(f"msgpack type 0x{t:02x} is not part of a code-object note")
    return obj()


def kernel_resources(lib_path: str) -> Dict[str, dict]:
    """{demangled kernel name: {vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch_bytes, lds_bytes, max_workgroup, waves_per_simd,
    workgroups_per_cu}} for every kernel of the library.  Occupancy: the 512-entry unified register file of a gfx950 SIMD in granules of
    8, at most 8 waves per SIMD; 160 KB of LDS per CU."""
    rows, mangled = [], []
    for elf in code_objects(lib_path):
        for name, ntype, desc in _notes(elf):
            if name != "AMDGPU" or ntype != 32:                        # NT_AMDGPU_METADATA
                continue
            md = _unpack_msgpack(desc)
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
        if is_own_kernel(name) and any(re.search(g, name) for g in gated) and (r["vgpr_spill"] or r["scratch_bytes"] or r["dynamic_stack"]):
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
        raise RuntimeError("kernel resource gate: no shipped kernel may spill vector registers or use scratch (k_fwd_cr4: cross-lane reads of a "
                           "register reloaded under a partial EXEC mask return stale lanes; everywhere else: a spill is a silent 2x):\n  " + "\n  ".join(bad))
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
