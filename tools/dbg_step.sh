for d in 0 0; do python tools/dbg_step.py 2>&1 | grep "TOTAL\|Error\|error" ; done
python -m pytest tests/test_nuts_step_free_gpu.py tests/test_nuts_spec_gpu.py -q -p no:cacheprovider 2>&1 | tail -4
