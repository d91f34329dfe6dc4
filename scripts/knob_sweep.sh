#!/bin/bash
# one-at-a-time sweep of the launch-rule knobs on the headline train step (bench.py --lean --train-only) -> gpurun_out/knob_sweep.log
cd $GRAFT_REPO_ROOT
run() {
  r=$(env "$@" timeout -k 10 120 python bench.py --lean --train-only --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])")
  echo "$* -> $r ms" | tee -a gpurun_out/knob_sweep.log
}
: > gpurun_out/knob_sweep.log
run CR_NOP=1
run CR_NOP=2
run CR_GRP_KSPLIT=2
run CR_GRP_KSPLIT=4
run CR_SPLITK_TARGET=192
run CR_SPLITK_TARGET=384
run CR_SPLITK_TARGET=512
run CR_CONV_DMA_MIN_TILES=64
run CR_CONV_DMA_MIN_TILES=256
run CR_IGEMM_KU8_BLOCKS=256
run CR_IGEMM_KU8_BLOCKS=512
run CR_IGEMM_KU_SMALL=2
run CR_IGEMM_KU_SMALL=8
run CR_BN_FUSE_ROWS=128
run CR_BN_FUSE_ROWS=512
run CR_BN_FUSE_ROWS=1024
run CR_WG_F32_TM=64
run CR_WG_PATCH_BLOCKS=3
run CR_CONV_PATCH_BLOCKS=3
run CR_CONV_PATCH_BLOCKS=4
run CR_ROI_BWD_T=256
run CR_CONV_BM64=0
run CR_CONV_KG2=0
