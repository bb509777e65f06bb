#!/bin/bash
# Development aid: rocprofv3 counter passes (one group per run; --pmc never combined with sys/hip tracing) over one
# scripts/kcase.py case.   scripts/pmc_probe.sh <case> <tag>   -> gpurun_out/pmc_<tag>/<group>/
case=$1; tag=$2; ngroups=${3:-99}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  [ $i -ge $ngroups ] && break
  timeout 150 rocprofv3 --kernel-trace --pmc $grp -d $out/g$i -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/scripts/kcase.py $case 3 > $out/g$i.log 2>&1
  i=$((i+1))
done < <(if [ -n "$4" ]; then cat "$4"; else cat <<'GROUPS'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GROUPS
fi)
python $GRAFT_REPO_ROOT/scripts/pmc_probe_report.py $out
