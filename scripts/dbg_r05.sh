cd $GRAFT_REPO_ROOT
python scripts/conv_layer_ab.py bf16 > gpurun_out/r05_ab_bf16_pc.json 2>gpurun_out/ab.err
CHORE_CONV_LDS_BF16=1 python scripts/conv_layer_ab.py bf16 > gpurun_out/r05_ab_bf16_lds.json 2>>gpurun_out/ab.err
python scripts/conv_layer_ab.py fp16x3 > gpurun_out/r05_ab_x3_pc.json 2>>gpurun_out/ab.err
CHORE_CONV_LDS=1 python scripts/conv_layer_ab.py fp16x3 > gpurun_out/r05_ab_x3_lds.json 2>>gpurun_out/ab.err
CHORE_NO_CONV_SMALL=1 python scripts/conv_layer_ab.py fp16x3 > gpurun_out/r05_ab_x3_nosmall.json 2>>gpurun_out/ab.err
CHORE_NO_CONV_SMALL=1 python scripts/conv_layer_ab.py bf16 > gpurun_out/r05_ab_bf16_nosmall.json 2>>gpurun_out/ab.err
tail -3 gpurun_out/ab.err
