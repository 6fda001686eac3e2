cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQC_DCACHE[A-Z_]*\|SQ_INSTS_SMEM\|SQ_WAIT_INST_LDS\|SQ_INST_CYCLES_VMEM[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST[A-Z_]*" gpurun_out/avail.txt | sort -u | tr '\n' ' '
echo
