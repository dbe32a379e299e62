#!/bin/bash
# CPU only: disassemble one kernel of the in-tree library (or of LIB=<path>) into /tmp/isa/<name>.s.  usage: tools/kisa.sh <mangled-name-regex>
set -e
LIB=${LIB:-$(dirname "$0")/../lidar_rt_amd/csrc/liblrt_hip.so}
mkdir -p /tmp/isa
python3 - "$LIB" <<'PY'
import struct,re,sys
b=open(sys.argv[1],'rb').read()
k=0
for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"),b):
    p0=m.start(); n=struct.unpack_from("<Q",b,p0+24)[0]; off=p0+32
    for i in range(n):
        o,sz,ts=struct.unpack_from("<QQQ",b,off); off+=24
        trip=b[off:off+ts].decode(); off+=ts
        if trip.startswith("hip") and sz>0:
            open(f"/tmp/isa/co{k}.elf","wb").write(b[p0+o:p0+o+sz]); k+=1
PY
for co in /tmp/isa/co*.elf; do
  for sym in $(/opt/rocm/lib/llvm/bin/llvm-readelf -s $co | awk '{print $8}' | grep -E "$1" | grep -v '\.kd$' | sort -u); do
    /opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$sym $co | sed 's/\/\/.*//' > /tmp/isa/$sym.s
    echo "/tmp/isa/$sym.s $(wc -l < /tmp/isa/$sym.s) lines, scratch ops: $(grep -c scratch_ /tmp/isa/$sym.s)"
  done
done
