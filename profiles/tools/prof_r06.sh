#!/bin/bash
# rocprofv3 passes of the timed workload (bench.py --headline-only): kernel stats, FETCH_SIZE, WRITE_SIZE, MFMA busy - separate runs;
# and the same three for the configs[1] shape (N=20000, full storage) as the headline of a second set
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o headline -- python3 bench.py --headline-only --steps 3 --warmup 1 > $O/stats.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- python3 bench.py --headline-only --steps 1 --warmup 0 > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- python3 bench.py --headline-only --steps 1 --warmup 0 > $O/write.log 2>&1
echo write done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- python3 bench.py --headline-only --steps 1 --warmup 0 > $O/mfma.log 2>&1
echo mfma done
S="--headline-only --order 20000 --lowest 8 --max-dim 80 --storage full"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_stats -o small -- python3 bench.py $S --steps 20 --warmup 3 > $O/s_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/s_fetch -o f -- python3 bench.py $S --steps 3 --warmup 0 > $O/s_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/s_write -o w -- python3 bench.py $S --steps 3 --warmup 0 > $O/s_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/s_mfma -o m -- python3 bench.py $S --steps 3 --warmup 0 > $O/s_mfma.log 2>&1
echo small done
du -sh $O
# the reference's matrix-free test operator generated in the symmetric sweep (profiles/tools/harness_apply.py at N=100000: 8 / 16 / 32 columns): kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/h_stats -o h -- python3 profiles/tools/harness_apply.py 100000 8,16,32 > $O/h_stats.log 2>&1
echo harness done
# PMC passes of the harness operator's sweeps (round 6: the one-variable polynomial; VALU and MFMA busy)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/h_mfma -o m -- python3 profiles/tools/harness_apply.py 100000 8,16,32 > $O/h_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/h_valu -o v -- python3 profiles/tools/harness_apply.py 100000 8,16,32 > $O/h_valu.log 2>&1 || true
echo harness pmc done
# configs[3] (N=200000 generalized, GJD; profiles/tools/gjd_timing.py: a warm-up solve and six timed ones): kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o c3 -- python3 profiles/tools/gjd_timing.py > $O/c3_stats.log 2>&1
echo configs3 done
