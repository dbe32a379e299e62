#!/usr/bin/env python3
"""profiles/<tag>_resources.md: registers, spills, scratch and LDS of the shipped kernels, read from the code objects of liblrt_hip.so (no GPU needed).
usage: python tools/write_resources_md.py r06"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
from lidar_rt_amd import build as lrt_build, resources

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
lib = os.path.join(REPO, "lidar_rt_amd", "csrc", "liblrt_hip.so")
res = resources.kernel_resources(lib)
own = {k: v for k, v in res.items() if resources.is_own_kernel(k)}
defs = sorted({k.split("<")[0] for k in own})
table = subprocess.run([sys.executable, "-m", "lidar_rt_amd.resources"], capture_output=True, text=True, cwd=REPO).stdout
rows = [l for l in table.splitlines() if l.startswith("|") and "rocprim::" not in l]
txt = f"""# {tag}: registers, spills, scratch and LDS of the shipped kernels (kernel sources `{lrt_build.source_hash()}`)

Read from the `NT_AMDGPU_METADATA` notes of the gfx950 code objects inside `lidar_rt_amd/csrc/liblrt_hip.so` by `lidar_rt_amd/resources.py`
(`python -m lidar_rt_amd.resources`; its own msgpack-subset and name reader: no `msgpack` module, no `c++filt`).  `lidar_rt_amd.build.build()` and
`__graft_entry__.build()` run the gate on every call and FAIL when ANY own kernel (`k_*`, `kc_*`) reports a spilled VGPR or private (scratch) memory;
`tests/test_resources.py` asserts the same on the shipped library without a GPU, and that the product holds at most 35 kernel definitions.
"Spilled SGPR" = parked in VGPR lanes with `v_writelane` / `v_readlane` (EXEC-independent, no memory): allowed.

The product library: **{len(defs)} kernel definitions, {len(own)} instantiations** (round 5: 47 / 62 with the retired generations, which now compile only into
`liblrt_hip_legacy.so`, `-DLRT_LEGACY`).  `k_fwd_cr4<true, 4, false>` (the production trace kernel) stays at 96 VGPRs = five workgroups per CU with 28.4 KB of LDS;
`k_bwd_prep2` sits AT its cap of 64 registers (two 1024-thread workgroups per CU): two attempts of this round to give it more work spilled and were rejected by the gate
(`profiles/{tag}_experiments.md`).

{chr(10).join(rows)}
"""
open(os.path.join(REPO, "profiles", f"{tag}_resources.md"), "w").write(txt)
print(f"profiles/{tag}_resources.md: {len(defs)} definitions, {len(own)} instantiations, sources {lrt_build.source_hash()}")
