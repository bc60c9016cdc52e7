#!/bin/bash
# Builds k_alnf variants forced below their natural register count (scratch spills) and compares their records with the shipped library's on the GPU
# (tools/spill_probe.py).  usage: bash tools/spill_probe.sh build   (here or on the GPU box: hipcc cross-compiles)   |   bash tools/spill_probe.sh run   (GPU box)
#   w5        14-row kernels at 5 waves per SIMD (96 registers: ~40 spilled)
#   w5noasm   the same without the inline assembly of the column step
#   v96       14-row kernels at their 4 waves per SIMD, capped at 96 registers (amdgpu_num_vgpr)
#   r64w4     64-row kernels at 4 waves per SIMD (128 registers: ~25 spilled)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/tksm_amd/csrc
if [ "$1" = build ]; then
  make -s -j4 -C $C
  mkdir -p $C/_spill
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$C"
  sed "s/constexpr int ALNF_WAVES = 4;/constexpr int ALNF_WAVES = 5;/" $C/kernels.hip > $C/_spill/kernels_w5.hip
  cp $C/_spill/kernels_w5.hip $C/_spill/kernels_w5noasm.hip
  cp $C/kernels.hip $C/_spill/kernels_v96.hip
  sed "s/__launch_bounds__(64, ROWS == 64 ? 3 : ALNF_WAVES) ALNF_VGPR_CAP void k_alnf/__launch_bounds__(64, 4) void k_alnf/" $C/kernels.hip > $C/_spill/kernels_r64w4.hip
  for v in w5 w5noasm v96 r64w4; do
    X=""; [ $v = w5noasm ] && X="-DTKSM_NO_INLINE_ASM"; [ $v = v96 ] && X="-DTKSM_ALNF_NUM_VGPR=96"
    /opt/rocm/bin/hipcc $F $X -c $C/_spill/kernels_$v.hip -o $C/_spill/kernels_$v.o 2>&1 | grep -v "warning\|^ \|^$" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/tksm_amd/libtksmseq_spill_$v.so $C/_spill/kernels_$v.o $C/_build/mdf_kernels.o $C/_build/api.o $C/_build/mdf_ops.o \
        $C/_build/models.o $C/_build/hostio.o $C/_build/sequencer_module.o $C/_build/mdf_modules.o -lz -ldl
  done
  python3 - <<PY
import sys; sys.path.insert(0, "$R")
import __graft_entry__ as g
for v in ("w5", "w5noasm", "v96", "r64w4"):
    for n, r in g.kernel_resources("$C/_spill/kernels_%s.o" % v).items():
        if "k_alnf" in n: print(v, n[8:36], "vgprs", r.get("vgpr_count"), "spilled", r.get("vgpr_spill_count"), "scratch bytes", r.get("private_segment_fixed_size"))
PY
else
  export SPILL_DUMP=/tmp/spill_ref_$$
  for v in "" _spill_w5 _spill_w5noasm _spill_v96 _spill_r64w4; do
    echo "== libtksmseq$v.so"
    TKSMSEQ_LIB=$R/tksm_amd/libtksmseq$v.so timeout -k 10 200 python3 $R/tools/spill_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
  done
  rm -f /tmp/spill_ref_$$*
fi
