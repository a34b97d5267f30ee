# Convenience targets; the recipes themselves live in crdmodel_amd/csrc/Makefile (hipcc, gfx950) and oracle/Makefile (gcc).
PY ?= python

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build
	$(PY) -m pytest tests -q -m gpu

bench: build
	$(PY) bench.py

clean:
	$(MAKE) -C crdmodel_amd/csrc clean
	$(MAKE) -C oracle clean

.PHONY: build test-cpu test-gpu bench clean
