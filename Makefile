# Thin wrappers around the Python entry points, with the parameters of the reference's targets
# (reference Makefile:17-35 `small`, :101-141 `kagome_36` / `pyrochlore_32` / `sk_32_1`).  The
# 16- and 18-site models of annealing_sign_problem_amd/models.json are diagonalised on the spot;
# the ground states of the large ones (the HDF5 files the reference's Makefile downloads, :143-153)
# are taken from DATA=<directory> or, for kagome_36 and pyrochlore_32, computed on the GPU.
PYTHON ?= python3
SEED ?= 435834
NOISE ?= 0
CUTOFF ?= 1e-6
ORDER ?= 2
NUMBER_SAMPLES ?= 1000
JOBS ?= 16
MODEL ?= heisenberg_kagome_16
SMALL_MODELS = heisenberg_kagome_16 heisenberg_kagome_18 j1j2_square_4x4 sk_16_1 sk_16_2 sk_16_3
DATA ?= physical_systems/data-large
OUT ?= experiments

.PHONY: all build test test-gpu bench small clusters kagome_36 pyrochlore_32 sk_32_1
all: build

build:
	$(PYTHON) __graft_entry__.py

test:
	$(PYTHON) -m pytest tests -x -q -m "not gpu"

test-gpu:
	$(PYTHON) -m pytest tests -x -q -m gpu

bench:
	$(PYTHON) bench.py

# `make small`: P(exact signs) vs number of sweeps on the full 16-site Hilbert spaces
small: $(SMALL_MODELS:%=$(OUT)/%.csv)

$(OUT)/%.csv:
	@mkdir -p $(OUT)
	$(PYTHON) -m annealing_sign_problem_amd.full_hilbert_space --model $* --seed $(SEED) \
		--output $@.wip --repetitions 1024 \
		--number-sweeps 100,200,400,800,1600,3200,6400,12800,25600,51200,102400,204800 && \
	mv $@.wip $@

# `make kagome_36`-style run (greedy only, extension order $(ORDER), global cutoff) on $(MODEL)
clusters:
	@mkdir -p $(OUT)/$(MODEL)/noise_$(NOISE)/cutoff_$(CUTOFF)
	$(PYTHON) -m annealing_sign_problem_amd.sampled_components --model $(MODEL) --seed $(SEED) \
		--output $(OUT)/$(MODEL)/noise_$(NOISE)/cutoff_$(CUTOFF)/$(MODEL).csv$(JOBID) \
		--order $(ORDER) --noise $(NOISE) --no-annealing --global-cutoff $(CUTOFF) \
		--number-samples $(NUMBER_SAMPLES) --jobs $(JOBS)

# The reference's large targets (Makefile:101-141): greedy only, order $(ORDER), cutoff $(CUTOFF),
# $(NUMBER_SAMPLES) sampled clusters (reference: 50000), on the model's own symmetry-adapted basis.
kagome_36: LARGE = heisenberg_kagome_36
pyrochlore_32: LARGE = heisenberg_pyrochlore_2x2x2
sk_32_1: LARGE = sk_32_1
# RANKS > 1: that many processes (clusters c mod RANKS, rank 0 writes the CSV; identical output):
# for a node with RANKS GPUs — drop the two environment variables and every rank binds its own
# GPU (RCCL).  On ONE GPU one process with JOBS threads is as fast (4096 clusters of kagome_36 in
# 27 s); as written the ranks share device 0 over gloo, a rehearsal.
RANKS ?= 1
ifeq ($(RANKS),1)
  LAUNCH = $(PYTHON) -m
else
  LAUNCH = ASP_DIST_BACKEND=gloo ASP_SINGLE_DEVICE=1 $(PYTHON) -m torch.distributed.run --nnodes=1 \
	--nproc-per-node $(RANKS) --master-addr 127.0.0.1 --master-port 29517 -m
endif
kagome_36 pyrochlore_32 sk_32_1:
	@mkdir -p $(DATA) $(OUT)/$(LARGE)/noise_$(NOISE)/cutoff_$(CUTOFF)
	@test -f $(DATA)/$(LARGE).h5 || $(MAKE) $(DATA)/$(LARGE).h5
	$(LAUNCH) annealing_sign_problem_amd.sampled_components --model $(LARGE) --hdf5 $(DATA)/$(LARGE).h5 \
		--seed $(SEED) --output $(OUT)/$(LARGE)/noise_$(NOISE)/cutoff_$(CUTOFF)/$(LARGE).csv$(JOBID) \
		--order $(ORDER) --noise $(NOISE) --no-annealing --global-cutoff $(CUTOFF) \
		--number-samples $(NUMBER_SAMPLES) --jobs $(JOBS)

# The ground states the reference downloads (Makefile:143-153, SpinED output) are regenerated on
# the GPU when absent: representatives, resident Hamiltonian and Lanczos of the whole symmetry
# sector (annealing_sign_problem_amd/sector_ed.py; heisenberg_kagome_36: 31.5 M representatives,
# about 10 s; sk_32_1: 6.0e8 states, matrix-free product, about 3 minutes and 10 GB of file).
$(DATA)/heisenberg_kagome_36.h5 $(DATA)/heisenberg_pyrochlore_2x2x2.h5 $(DATA)/sk_32_1.h5:
	@mkdir -p $(DATA)
	$(PYTHON) -m annealing_sign_problem_amd.sector_ed --model $(basename $(notdir $@)) --output $@ --tol 1e-8
