python tools/conv_bench.py 2>&1 | grep -v "^X"
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not ten_thousand" 2>&1 | tail -3
