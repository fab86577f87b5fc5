# oracle/ref.mk -- builds what of the REAL reference compiles from its own few source files, where they lie under /root/reference, into
# oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).  Test infrastructure only; nothing is copied into the repo.
#   libminilzo.so : Application/src/ProcessedVideo/lzo/minilzo.c, unmodified -- lzo1x_1_compress / lzo1x_decompress as pv::Frame::serialize
#                   and pv::Frame::read_from call them (ProcessedVideo/pv.cpp:331,738)
#   make -f oracle/ref.mk          (from the repo root; a no-op with a message when the reference tree is absent)
REF ?= /root/reference
LZO = $(REF)/Application/src/ProcessedVideo/lzo
CC ?= gcc
OUT = oracle/_ref

all:
	@if [ -f $(LZO)/minilzo.c ]; then $(MAKE) -f oracle/ref.mk $(OUT)/libminilzo.so; else echo "oracle/ref.mk: no reference tree at $(REF): nothing to build"; fi

$(OUT)/libminilzo.so: $(LZO)/minilzo.c $(LZO)/minilzo.h $(LZO)/lzoconf.h $(LZO)/lzodefs.h
	mkdir -p $(OUT)
	$(CC) -O2 -fPIC -shared -I$(LZO) -o $@ $(LZO)/minilzo.c
