#!/bin/bash
O=gpurun_out/r3_43; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest exit $?" | tee -a $O/ab.txt; tail -2 $O/pytest.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke exit $?" | tee -a $O/ab.txt
