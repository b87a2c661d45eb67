for s in 6 8 12; do echo "split $s"; LEGO_TN_SPLIT=$s python tools/bert_shapes_bench.py 2>&1 | grep "dW"; done
