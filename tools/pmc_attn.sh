#!/bin/bash
# Where the attention kernel's wave-cycles go: SQ wait / active / LDS-conflict counters (separate rocprofv3 --pmc passes, no trace domains)
# over the kernel micro-benchmark, for the library in place and optionally a measurement variant (tools/attn_ablate.sh).
#   bash tools/pmc_attn.sh <tag> [variant.so ...]          (on the GPU box; writes gpurun_out/<tag>_pmc_attn.txt)
tag=${1:-attn}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_pmc_attn.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > /tmp/avail.txt
cp /tmp/avail.txt $R/gpurun_out/${tag}_pmc_avail.txt
want="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC \
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM \
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64 SQ_VALU_MFMA_BUSY_CYCLES"
have=""
for c in $want; do grep -qx $c /tmp/avail.txt && have="$have $c"; done
echo "# counters: $have" > $out
cp $R/foundpose_amd/lib/libfoundpose_amd.so /tmp/lib_inplace.so
for lib in inplace "$@"; do
  [ $lib = inplace ] || cp $R/foundpose_amd/lib/$lib $R/foundpose_amd/lib/libfoundpose_amd.so
  echo "== library: $lib" >> $out
  set -- $have
  pass=0
  while [ $# -gt 0 ]; do
    grp="$1 $2 $3 $4"; shift; shift; shift; shift
    d=/tmp/pmca_${lib}_$pass; rm -rf $d; pass=$((pass+1))
    BENCH_ATTN_VARIANTS=0 timeout 200 rocprofv3 --pmc $grp -d $d -o r -- python $R/tools/bench_kernels.py attn > /dev/null 2>&1 || echo "pass '$grp' failed" >> $out
    db=$(find $d -name "*.db" 2>/dev/null | head -1)
    [ -n "$db" ] && python $R/tools/pmc_summary.py $db attn_bf16 | awk '{print "   ", $(NF-2), $(NF-1), $NF}' >> $out
  done
  cp /tmp/lib_inplace.so $R/foundpose_amd/lib/libfoundpose_amd.so
done
cat $out
