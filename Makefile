# Convenience targets; every sub-Makefile also works on its own.
.PHONY: all lib oracle cpp tools test clean
all: lib oracle cpp tools
lib:
	$(MAKE) -C x-search_amd
oracle:
	$(MAKE) -C oracle
cpp: lib
	$(MAKE) -C tests/cpp
tools: lib
	$(MAKE) -C tools
test: all
	python -m pytest tests -q -m "not gpu"
clean:
	$(MAKE) -C x-search_amd clean
	$(MAKE) -C oracle clean
	rm -rf tests/cpp/build tools/build tests/cpu_model/build scripts/microbench/build
