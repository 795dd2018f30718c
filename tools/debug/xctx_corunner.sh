mkdir -p gpurun_out/xctx4
export BNHIP_LIB=$PWD/birdnet-go_amd/lib/libbnhip_slp.so BNHIP_DEBUG_SYNC_AFTER=melband TRIALS=300 SAME=1
S=tools/debug/ctx_sig.py
run() { for r in 1 2 3; do timeout 200 python $S 2>&1 | grep -E "bad trials|Error|error" >> gpurun_out/xctx4/$1.txt; done; }
run A_full
BNHIP_DEBUG_CTX1_SKIP=pw_gemm run B_skip_pw
BNHIP_DEBUG_CTX1_SKIP=expand_dw run C_skip_expdw
BNHIP_DEBUG_CTX1_SKIP=se run D_skip_se
BNHIP_DEBUG_CTX1_SKIP=dwconv,mean run E_skip_dw
BNHIP_BF16X3=0 run F_nobx3
BNHIP_DEBUG_CTX1_SKIP=pw_gemm,expand_dw run G_skip_mfma
tail -n 4 gpurun_out/xctx4/*.txt
