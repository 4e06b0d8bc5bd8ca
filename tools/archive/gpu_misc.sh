for i in 1 2 3 4 5; do timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -1; done
RDR_WORKERS=1 timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -1
RDR_NO_OVERLAP=1 timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
