#!/bin/bash
# population run: the whole bench (fp32 parity leg, batch-1 legs, both training legs) against a find-db + kernel cache under gpurun_out/
cd "$(dirname "$0")/.."
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_userdb
export MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen_userdb/cache
mkdir -p $MIOPEN_CUSTOM_CACHE_DIR
timeout 1500 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/r06_db_bench.json 2> gpurun_out/r06_db_bench.err
grep "train step\|parity" gpurun_out/r06_db_bench.err
echo "== second process"
timeout 1500 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/r06_db_bench2.json 2> gpurun_out/r06_db_bench2.err
grep "train step\|parity\|hpatches" gpurun_out/r06_db_bench2.err
ls -la gpurun_out/miopen_userdb gpurun_out/miopen_userdb/cache
