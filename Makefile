# Build recipe for the host library, the oracle, the HIP library and (in the build
# container only) the reference pin under oracle/_ref.  No cmake/ninja needed.
CC      ?= gcc
HIPCC   ?= /opt/rocm/bin/hipcc
CFLAGS  ?= -O2 -g -std=gnu11 -fPIC -Wall -Wextra -Wno-unused-parameter
REF     ?= /root/reference

HOST_SRC = marlin_amd/host/mdict.c marlin_amd/host/mshard.c marlin_amd/host/manalyzer.c marlin_amd/host/mhighlight.c \
           marlin_amd/host/mquery.c marlin_amd/host/mplan.c marlin_amd/host/mbridge.c
HOST_HDR = $(wildcard marlin_amd/host/*.h) include/marlin_gpu.h
ORC_SRC  = oracle/orc_squery.c oracle/orc_sort.c oracle/orc_merge.c
GPU_SRC  = $(wildcard marlin_amd/csrc/*.hip)
GPU_HDR  = $(wildcard marlin_amd/csrc/*.h) include/marlin_gpu.h

all: gpu host oracle

gpu: marlin_amd/lib/libmarlin_gpu.so
host: marlin_amd/lib/libmarlin_host.so
oracle: oracle/liborc.so

marlin_amd/lib/libmarlin_gpu.so: $(GPU_SRC) $(GPU_HDR)
	@mkdir -p marlin_amd/lib
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Imarlin_amd/csrc \
	    -o $@ $(GPU_SRC) -L/opt/rocm/lib -lrccl

# the same library with the checked accessors of the rank / doc-processing kernels compiled in (mgpu_kernels.hip, CK):
# `make gpu-checked` overwrites libmarlin_gpu.so; run the GPU suite on it, then `make -B gpu` for the product build
gpu-checked: $(GPU_SRC) $(GPU_HDR)
	@mkdir -p marlin_amd/lib
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMGPU_CHECKED -Iinclude -Imarlin_amd/csrc \
	    -o marlin_amd/lib/libmarlin_gpu.so $(GPU_SRC) -L/opt/rocm/lib -lrccl

marlin_amd/lib/libmarlin_host.so: $(HOST_SRC) $(HOST_HDR) marlin_amd/lib/libmarlin_gpu.so
	@mkdir -p marlin_amd/lib
	$(CC) $(CFLAGS) -shared -Iinclude -o $@ $(HOST_SRC) -Lmarlin_amd/lib -lmarlin_gpu \
	    -Wl,-rpath,'$$ORIGIN' -lm -lpthread

oracle/liborc.so: $(ORC_SRC) oracle/orc.h $(HOST_HDR) marlin_amd/lib/libmarlin_host.so
	$(CC) $(CFLAGS) -shared -o $@ $(ORC_SRC) -Lmarlin_amd/lib -lmarlin_host \
	    -Wl,-rpath,'$$ORIGIN/../marlin_amd/lib' -lm -lpthread

# The reference's own bitmap / sort / hashtable / farmhash sources, compiled unmodified from
# where they lie.  Only possible where /root/reference exists (the build container).
ref: oracle/_ref/libmarlin_ref.so
oracle/_ref/libmarlin_ref.so:
	@mkdir -p oracle/_ref
	$(CC) -std=gnu99 -O2 -fPIC -shared -D_GNU_SOURCE -DH2O_USE_LIBUV=0 \
	    -I$(REF)/src/inc -I$(REF)/src/main -I$(REF)/src/utils \
	    -I$(REF)/deps/jansson/src -I$(REF)/deps/jansson/android -I$(REF)/deps/h2o/include \
	    $(REF)/src/main/bmap.c $(REF)/src/main/cont.c $(REF)/src/main/sort.c \
	    $(REF)/src/main/hashtable.c $(REF)/src/utils/farmhash-c.c -o $@

# The reference's index-writing code (sindex.c, compiled where it lies by oracle/ref_seam.c's #include) over the
# reference's own LMDB (mdb.c, midl.c): writes docid2data blobs and num_dbi entries for tests/golden/docdata_cases.json.
# Symbols of the reference the program never calls stay unresolved.  Build container only.
REF_INC = -I$(REF)/src/inc -I$(REF)/src/main -I$(REF)/src/utils -I$(REF)/src/analyzer \
          -I$(REF)/deps/jansson/src -I$(REF)/deps/jansson/android -I$(REF)/deps/h2o/include -I$(REF)/deps/lmdb \
          -I$(REF)/deps/utf8proc -I$(REF)/deps
ref-seam: oracle/_ref/ref_seam
oracle/_ref/ref_seam: oracle/ref_seam.c
	@mkdir -p oracle/_ref
	$(CC) -std=gnu99 -O1 -g -w -D_GNU_SOURCE -DH2O_USE_LIBUV=0 $(REF_INC) oracle/ref_seam.c \
	    $(REF)/deps/lmdb/mdb.c $(REF)/deps/lmdb/midl.c $(REF)/src/utils/mlog.c -Wl,--unresolved-symbols=ignore-all -o $@ -lpthread -lm

# The reference's own shard query path on a shard its own indexing code built (oracle/ref_query.c): sindex.c, aggs.c and
# metric-aggs.c are compiled by inclusion, everything else of the reference as it lies.  No -DNDEBUG: farmhash32 then applies its
# debug tweak (farmhash-c.c:287-293), the variant m_facet_id restates.  Build container only.
REF_M = $(REF)/src/main
REF_QSRC = $(REF_M)/docrank.c $(REF_M)/filter_apply.c $(REF_M)/filter.c $(REF_M)/sort.c $(REF_M)/hashtable.c \
           $(REF_M)/bmap.c $(REF_M)/cont.c $(REF_M)/mbmap.c $(REF_M)/dtrie.c $(REF_M)/query.c $(REF_M)/highlight.c $(REF_M)/shard.c \
           $(REF_M)/sdata.c $(REF_M)/mapping.c $(REF_M)/workers.c $(REF_M)/debug.c $(REF)/src/utils/utils.c \
           $(REF)/src/utils/farmhash-c.c $(REF)/src/utils/mlog.c $(REF)/src/utils/threadpool.c $(REF)/src/utils/lz4.c \
           $(REF)/src/analyzer/analyzer.c $(REF)/src/analyzer/default_analyzer.c $(REF)/src/analyzer/whitespace_analyzer.c \
           $(REF)/src/analyzer/word.c $(REF)/deps/utf8proc/utf8proc.c $(REF)/deps/lmdb/mdb.c $(REF)/deps/lmdb/midl.c
REF_QFLAGS = -std=gnu99 -O2 -g -w -D_GNU_SOURCE -DH2O_USE_LIBUV=0 -DH20_USE_EPOLL $(REF_INC)
ref-query: oracle/_ref/ref_query
oracle/_ref/ref_query: oracle/ref_query.c
	@mkdir -p oracle/_ref
	$(CC) $(REF_QFLAGS) oracle/ref_query.c $(REF_M)/squery.c $(REF_QSRC) -Wl,--unresolved-symbols=ignore-all -o $@ -lpthread -lm -lz

# The same program with shim/squery_gpu.c IN PLACE OF the reference's squery.c: the reference's indexing, dtrie, LMDB and
# query structures on the host, the shard query itself on the GPU through the seam.  Runs on the GPU box (oracle/_ref/
# travels); tests/test_gpu_shim_seam.py compares it with ref_query.
ref-query-gpu: oracle/_ref/ref_query_gpu
oracle/_ref/ref_query_gpu: oracle/ref_query.c shim/_build/squery_gpu.o marlin_amd/lib/libmarlin_host.so
	@mkdir -p oracle/_ref
	$(CC) $(REF_QFLAGS) -DREF_QUERY_GPU oracle/ref_query.c shim/_build/squery_gpu.o $(REF_QSRC) -Wl,--unresolved-symbols=ignore-all \
	    -Lmarlin_amd/lib -lmarlin_host -lmarlin_gpu -Wl,-rpath,'$$ORIGIN/../../marlin_amd/lib' -o $@ -lpthread -lm -lz

# The drop-in replacement for the reference's squery.c, compiled against the reference's own headers (same include
# paths as `ref`).  Build container only; the object never travels and nothing links it here: the check is that it
# compiles without warnings and exports execute_squery / sqresult_free (tests/test_shim.py).
shim: shim/_build/squery_gpu.o shim/_build/query_gpu.o
shim/_build/query_gpu.o: shim/query_gpu.c
	@mkdir -p shim/_build
	$(CC) -std=gnu99 -O2 -fPIC -w -D_GNU_SOURCE -DH2O_USE_LIBUV=0 $(REF_INC) -c shim/query_gpu.c -o $@
shim/_build/squery_gpu.o: shim/squery_gpu.c $(HOST_HDR)
	@mkdir -p shim/_build
	$(CC) -std=gnu99 -O2 -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -Werror -D_GNU_SOURCE -DH2O_USE_LIBUV=0 \
	    -I$(REF)/src/inc -I$(REF)/src/main -I$(REF)/src/utils -I$(REF)/src/analyzer \
	    -I$(REF)/deps/jansson/src -I$(REF)/deps/jansson/android -I$(REF)/deps/h2o/include -I$(REF)/deps/lmdb \
	    -I$(REF)/deps/utf8proc -I$(REF)/deps -Iinclude -Imarlin_amd/host -c shim/squery_gpu.c -o $@

clean:
	rm -f marlin_amd/lib/*.so oracle/liborc.so oracle/_ref/*.so shim/_build/*.o

.PHONY: all gpu gpu-checked host oracle ref ref-seam ref-query ref-query-gpu shim clean
