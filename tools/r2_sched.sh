#!/bin/bash
t() { timeout 300 python -m pytest tests/test_model_gpu.py -x -q -s -k "bf16_precision" 2>&1 | grep -E "max\|" | tail -1; }
echo "default: $(t)"
echo "tail>=96: $(MGNNS_FUSED_TAIL_BF16_MIN_BATCH=96 t)"
echo "lgcn unfused: $(MGNNS_FUSED_LABEL_GCN=0 t)"
echo "head unfused: $(MGNNS_FUSED_HEAD=0 t)"
echo "cluster off: $(MGNNS_LABEL_TAIL_CLUSTER=0 t)"
echo "all old: $(MGNNS_FUSED_TAIL_BF16_MIN_BATCH=96 MGNNS_FUSED_LABEL_GCN=0 MGNNS_FUSED_HEAD=0 t)"
echo "lstm f32: $(MGNNS_LSTM_REC=f32 t)"
