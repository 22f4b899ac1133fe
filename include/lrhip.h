/*
 * lrhip.h - C ABI of liblrhip.so, the MI355X (gfx950) DSP block engine behind LuaRadio's block API.
 *
 * This is the drop-in boundary: what a LuaJIT `ffi.cdef` (see INTEGRATION.md and lua/radio/) binds in
 * place of the VOLK / liquid-dsp / FFTW3f entry points the reference's blocks call from process().
 * Plain pointers and sizes only; no C++ or torch types.  All paths below are relative to /root/reference.
 *
 * Conventions (liquid-dsp style, as used by the reference at radio/blocks/signal/firfilter.lua:167-226):
 *   - `T *q = lrhip_X_create(params)` returns NULL on error (the reference's blocks then raise
 *     `error("Creating ... object")`, firfilter.lua:199-201); `lrhip_strerror()` gives the text.
 *   - `lrhip_stage_execute(q, in, n, out, cap)` is one block's process(): host pointers in, host pointers
 *     out, returns the number of output samples written (>= 0) or < 0 on error.  All cross-call state
 *     (FIR history, rotator phase, downsampler index, discriminator previous sample, IIR state) lives in
 *     the stage object on the device, so arbitrary chunking gives identical sample values for every block on
 *     its own in direct form (the property tests/jigs.lua:213-250 pins with one-sample chunks).  Overlap-save
 *     filters and the fused forms a DEFAULT chain takes (listed at lrhip_chain_create) agree across chunkings
 *     to Float32 rounding (<= 1e-6 / 2e-7), not bit for bit; lrhip_chain_create_ex(.., LRHIP_CHAIN_EXACT)
 *     restores the bits.
 *   - `lrhip_stage_destroy(q)` is bound with ffi.gc.
 *   - Sample layouts are the reference's: ComplexFloat32 = struct{float real, imag} (8 B, interleaved,
 *     radio/types/complexfloat32.lua:19-24), Float32 = struct{float value} (4 B, radio/types/float32.lua:17-21).
 *   - Nothing here throws, aborts or prints.  Single-threaded use per process, like a LuaRadio block
 *     (one process per block after fork(): radio/core/composite.lua:569); lrhip_init() must first be
 *     called AFTER fork (it creates the HIP context lazily on first use).
 */
#ifndef LRHIP_H
#define LRHIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lrhip_stage lrhip_stage_t;   /* one block's device state + kernels */
typedef struct lrhip_chain lrhip_chain_t;   /* a linear run of stages with device-resident edges */

/* ---- runtime -------------------------------------------------------------------------------------- */
/* Select the device and create the stream.  Replaces platform.load()/feature detection for a native
 * library (radio/core/platform.lua:277-299).  device < 0 => current device.  Idempotent. */
int lrhip_init(int device);
/* Text of the last error on this thread ("" if none). */
const char *lrhip_strerror(void);
/* Number of visible HIP devices (for the fan-out scheduler), or < 0 on error. */
int lrhip_device_count(void);
/* The device this process's library is bound to (what lrhip_init chose; the dst_device / src_device of lrhip_peer_copy), -1 before the
 * first lrhip_init / stage creation.  One device per process: after a successful lrhip_init(d) a later lrhip_init(d2) with d2 >= 0 and
 * d2 != d fails - LuaRadio runs one process per block (radio/core/composite.lua:569), so "a branch per GPU" is "a device per process". */
int lrhip_device(void);
/* Launch all subsequent work on an externally owned hipStream_t (NULL => the library's own non-blocking stream; HIP's explicit handles
 * hipStreamLegacy = (hipStream_t)1 and hipStreamPerThread = (hipStream_t)2 are accepted - a host that keeps its vectors on the default
 * stream passes 1).  Work already queued on the previous stream is ordered before whatever follows on the new one. */
int lrhip_set_stream(void *hip_stream);
/* Block until everything queued by this library has finished. */
int lrhip_synchronize(void);
/* Library version string. */
const char *lrhip_version(void);

/* ---- stage constructors ---------------------------------------------------------------------------- */
/* FIRFilterBlock (firfilter.lua:43-74 instantiate, :90-163 / :230-305 dot-product form, :320-398 overlap-save
 * framing).  taps: ntaps floats (taps_complex = 0) or ntaps {re,im} pairs (taps_complex = 1), in the
 * reference's natural order h[0..M-1] (the library reverses them, :234-238).  input_complex selects
 * ComplexFloat32 or Float32 samples; complex taps require complex input (:69-74).
 * decim >= 1 fuses a following DownsamplerBlock(decim) (radio/composites/decimator.lua:37-39): only every
 * decim-th filter output is computed and emitted, with the downsampler's carried phase index.
 * use_fft: 0 = direct form (bit-identical to the fmaf chain in the reference's tap order);
 *          1 = overlap-save as the reference runs it (:320-398): only whole L = N-M+1 blocks are emitted,
 *              N = 2^floor(log2(8M)), the tail is retained; arithmetic by the fused FFT kernel for 32 <= M <= 8192;
 *          2 = overlap-save arithmetic, 32 <= M <= 8192, with sample-exact emission (every call returns one output per input, like
 *              the direct form) - the fast path: the fused 1024-point kernel to 512 taps, 4096-point blocks above (one per wave as
 *              64 x 64 - real or complex taps on a ComplexFloat32 stream, real taps on a Float32 stream; two partitions per launch
 *              above 2 049 taps, two launches above 4 097), <= 1e-6 of the exact result whichever kernel the launch size selects;
 *          3 = automatic: 2 when the filter qualifies and has at least 48 taps (where the FFT form is the faster one on
 *              MI355X), else 0.  Outside the stated tap range 1 and 2 keep their emission rule and use the direct arithmetic. */
lrhip_stage_t *lrhip_fir_create(const float *taps, unsigned ntaps, int taps_complex, int input_complex,
                                unsigned decim, int use_fft);
/* FrequencyTranslatorBlock (radio/blocks/signal/frequencytranslator.lua:26-53, :93-110):
 * y[n] = x[n] * exp(j*omega*n), omega = 2*pi*offset/rate computed by the caller as at :95. */
lrhip_stage_t *lrhip_rotator_create(double omega);
/* DownsamplerBlock (radio/blocks/signal/downsampler.lua:29-56). elem_size = 8 (ComplexFloat32) or 4 (Float32). */
lrhip_stage_t *lrhip_downsampler_create(unsigned factor, int elem_size);
/* FrequencyDiscriminatorBlock (radio/blocks/signal/frequencydiscriminator.lua:25-38, :48-88).
 * gain = 2*pi*modulation_index (:28).  ComplexFloat32 in, Float32 out. */
lrhip_stage_t *lrhip_fmdiscrim_create(double gain);
/* IIRFilterBlock (radio/blocks/signal/iirfilter.lua:39-61, :113-181): b[nb] feed-forward, a[na] feedback
 * (a[0] divides).  SinglepoleLowpassFilterBlock / FMDeemphasisFilterBlock compute b,a on the host
 * (singlepolelowpassfilter.lua:55-67) and call this with nb = na = 2. */
lrhip_stage_t *lrhip_iir_create(const float *b, unsigned nb, const float *a, unsigned na, int input_complex);
/* spectrum_utils.PSD (radio/utilities/spectrum_utils.lua:522-561, :585-640) over frames of n samples:
 * window multiply -> n-point DFT -> |X|^2/scale -> optional 10*log10 -> optional fftshift (:654-667).
 * window: n floats (the caller builds the periodic window as at :547); scale = sample_rate * sum(w^2) (:597).
 * Input n_in must be a multiple of n; output is n_in Float32 values.  n must be a power of two, 8..4096. */
lrhip_stage_t *lrhip_psd_create(unsigned n, const float *window, double scale, int logarithmic,
                                int input_complex, int fftshift);
/* spectrum_utils.DFT / IDFT (spectrum_utils.lua:25-113, :259-349) over frames of n samples.
 * inverse != 0 applies the 1/n normalisation (:335-338).  real_side: for forward transforms a Float32 input,
 * for inverse transforms a Float32 output (real part, :499-503).  Complex side is ComplexFloat32. */
lrhip_stage_t *lrhip_dft_create(unsigned n, int inverse, int real_side);

/* IQFileSource / RealFileSource sample conversion (radio/blocks/sources/iqfile.lua:99-113, realfile.lua:99-110,
 * formats table radio/utilities/format_utils.lua:82-97): raw file samples -> (value - offset)/scale.
 * format: "u8","s8","u16le","u16be","s16le","s16be","u32le","u32be","s32le","s32be","f32le","f32be","f64le","f64be".
 * complex_out != 0: input is interleaved I/Q (2 raw scalars per sample), output ComplexFloat32; else Float32.
 * Input samples are the RAW file records (lrhip_stage_input_size() bytes each), so the bytes read from the file
 * are handed over unchanged and converted on the device. */
lrhip_stage_t *lrhip_format_convert_create(const char *format, int complex_out);
/* The sink direction (radio/blocks/sinks/iqfile.lua:68-85, realfile.lua): ComplexFloat32 / Float32 samples -> raw file records
 * raw = x*scale + offset stored into the format's type (C conversion, i.e. truncation for the integer formats), byte-swapped
 * for the big-endian formats.  Output samples are lrhip_stage_output_size() bytes each. */
lrhip_stage_t *lrhip_format_pack_create(const char *format, int complex_in);

/* Two-input element-wise blocks: op = "multiply" (radio/blocks/signal/multiply.lua:43-76), "multiplyconjugate"
 * (multiplyconjugate.lua:41-59, complex only), "add" (add.lua), "subtract" (subtract.lua), "floattocomplex" (floattocomplex.lua: two Float32 inputs ->
 * ComplexFloat32 (in1, in2); input_complex ignored).  Replaces
 * volk_32fc_x2_multiply_32fc_a / volk_32f_x2_multiply_32f_a / volk_32fc_x2_multiply_conjugate_32fc_a.
 * Executed with lrhip_stage_execute2*(). */
lrhip_stage_t *lrhip_binary_create(const char *op, int input_complex);

/* MultiplyConstantBlock (radio/blocks/signal/multiplyconstant.lua:28-71): real constant (constant_complex = 0, `im`
 * ignored) on Float32 or ComplexFloat32 input, or a complex constant on ComplexFloat32 input only (:42-44). */
lrhip_stage_t *lrhip_multiply_constant_create(float re, float im, int constant_complex, int input_complex);
/* UpsamplerBlock (radio/blocks/signal/upsampler.lua:26-53): zero-stuffing by `factor`. elem_size = 8 or 4. */
lrhip_stage_t *lrhip_upsampler_create(unsigned factor, int elem_size);

/* FrequencyModulatorBlock (radio/blocks/signal/frequencymodulator.lua:24-90; replaces freqmod_create / freqmod_modulate_block
 * :33-51): Float32 in, ComplexFloat32 out = exp(j * running phase), phase += 2*pi*modulation_index*x[n]. */
lrhip_stage_t *lrhip_fmmod_create(double modulation_index);

/* AGCBlock (radio/blocks/signal/agc.lua:25-96).  power_alpha = 1/(1 + power_tau*rate), gain_alpha = 1/(1 + gain_tau*rate)
 * (:46-49), target / threshold LINEAR power (10^(dB/10), :53-55).  Both recurrences run as parallel scans in double. */
lrhip_stage_t *lrhip_agc_create(double power_alpha, double gain_alpha, double target, double threshold, int input_complex);

/* PowerSquelchBlock (radio/blocks/signal/powersquelch.lua:26-80): alpha = 1/(1 + tau*rate), threshold LINEAR power; samples
 * pass while the running average power is at or above the threshold, else zeros. */
lrhip_stage_t *lrhip_powersquelch_create(double alpha, double threshold, int input_complex);

/* One-input element-wise blocks. op: "complexmagnitude", "complexphase", "complextoreal", "complextoimag",
 * "complexconjugate" (ComplexFloat32 in), "realtocomplex", "absolutevalue" (Float32 in), and "addconstant"
 * (radio/blocks/signal/addconstant.lua:26-75: constant (re, im); constant_complex / input_complex as for
 * lrhip_multiply_constant_create).  For the ops with a fixed input type input_complex is ignored. */
lrhip_stage_t *lrhip_unary_create(const char *op, float re, float im, int constant_complex, int input_complex);
/* DelayBlock (radio/blocks/signal/delay.lua:26-72): delay by num_samples (> 0), zero initial state. elem_size 8 or 4. */
lrhip_stage_t *lrhip_delay_create(unsigned num_samples, int elem_size);
/* HilbertTransformBlock (radio/blocks/signal/hilberttransform.lua:25-37, :100-160): Float32 in, ComplexFloat32 out =
 * (input delayed by (M-1)/2, input filtered by the M Hilbert taps).  taps: M (odd) floats from fir_hilbert_transform. */
lrhip_stage_t *lrhip_hilbert_create(const float *taps, unsigned ntaps);

/* Welch / Bartlett averaged power spectrum, the arithmetic of GnuplotSpectrumSink / GnuplotWaterfallSink
 * (radio/blocks/sinks/gnuplotspectrum.lua:140-186): frames of n samples every (n - overlap) samples, each frame's PSD
 * (arguments as lrhip_psd_create, spectrum fftshifted) accumulated on the device.  The stage is a sink:
 * lrhip_stage_execute*() consumes the chunk and returns 0 outputs; partial frames are carried to the next call.
 * lrhip_welch_read() writes the n-point average (sum / frames) to avg_host and returns the number of frames averaged
 * (0: nothing accumulated, avg_host untouched); reset != 0 clears the accumulator (gnuplotspectrum.lua:189-191). */
lrhip_stage_t *lrhip_welch_create(unsigned n, const float *window, double scale, int logarithmic, int input_complex, unsigned overlap);
long lrhip_welch_read(lrhip_stage_t *q, float *avg_host, int reset);

/* Critically sampled K-channel analysis filterbank (BASELINE.json configs[4]; not a block of the reference: defined as
 * K parallel chains FrequencyTranslatorBlock(-c*fs/K) -> FIRFilterBlock(taps) -> DownsamplerBlock(K), c = 0..K-1).
 * ComplexFloat32 in; one output frame of K ComplexFloat32 values (channel-major within the frame) per K input
 * samples, with the downsampler's carried index.  Computed as one dense GEMM on the f32 matrix cores.
 * nchannels in {32, 64}; ntaps a multiple of 32. */
lrhip_stage_t *lrhip_channelizer_create(const float *taps, unsigned ntaps, unsigned nchannels);

void lrhip_stage_destroy(lrhip_stage_t *q);
/* Back to the just-created state (zero history, phase 0, index 0). */
int lrhip_stage_reset(lrhip_stage_t *q);
/* Bytes per input / output sample (8 or 4). */
int lrhip_stage_input_size(const lrhip_stage_t *q);
int lrhip_stage_output_size(const lrhip_stage_t *q);
/* Upper bound on the samples execute() will write for n_in inputs, so the caller can
 * `self.out:resize()` first (firfilter.lua:130). */
unsigned long lrhip_stage_max_output(const lrhip_stage_t *q, unsigned long n_in);

/* ---- execution --------------------------------------------------------------------------------------- */
/* One block's process() with host buffers: H2D through a pinned ring, kernels, D2H, wait.
 * Returns the number of output samples written, or < 0. */
long lrhip_stage_execute(lrhip_stage_t *q, const void *in_host, unsigned long n_in,
                         void *out_host, unsigned long out_capacity);
/* Same, with device pointers, asynchronous on the library stream (no copies, no wait).  The output count is
 * known on the host without a device round-trip.  This is what chains and bench.py use. */
long lrhip_stage_execute_device(lrhip_stage_t *q, const void *in_dev, unsigned long n_in,
                                void *out_dev, unsigned long out_capacity);

/* Two-input variants for lrhip_binary_create() stages (both inputs n_in samples). */
long lrhip_stage_execute2(lrhip_stage_t *q, const void *in1_host, const void *in2_host, unsigned long n_in,
                          void *out_host, unsigned long out_capacity);
long lrhip_stage_execute2_device(lrhip_stage_t *q, const void *in1_dev, const void *in2_dev, unsigned long n_in,
                                 void *out_dev, unsigned long out_capacity);

/* ---- chains: several blocks, device-resident edges ------------------------------------------------------ */
/* A maximal linear run of device-capable blocks collapsed into one object (what CompositeBlock's
 * _prepare_to_run would build, radio/core/composite.lua:426): one H2D at the head, one D2H at the tail,
 * intermediate vectors never leave HBM.  The chain borrows the stages (caller keeps ownership) and fuses
 * adjacent stages where a fused kernel exists (rotator -> FIR -> downsampler, FIR -> downsampler, ... -> discriminator, the 1/5-rate
 * audio tail of the FM receivers).  A chain gives the values of its blocks run one by one - the same bits, with FIVE stated exceptions
 * (each a different rounding of the same mathematics, none above 1e-6 of the exact result):
 *   1. overlap-save filters (Float32 FFT arithmetic: the blocks of a chunk fall where the chunk starts, <= 1e-6);
 *   2. the polyphase audio tail (FIR -> single-pole IIR -> downsampler as one decimating filter, ~2e-8 RMS);
 *   3. a frequency translator fused in front of a filter whose output only the discriminator sees: a tile's window is rotated relative to
 *      its first sample, which leaves the angles unchanged to Float32 rounding of the filter outputs;
 *   4. the FM receiver in ONE launch (kernels_rx.h; the default for the tuner + discriminator + audio-tail shape of
 *      examples/rtlsdr_wbfm_mono.lua:12-17 whose de-emphasis pole q at the audio rate satisfies q^75 <= 2^-25): the recurrence of every
 *      workgroup run restarts from a zero state 75 audio samples early and the runs fall with the chunk length, so the audio depends on
 *      the chunking and on the grid to <= 2e-7 (tests/test_gpu_rx.py), and time partitions agree with the single stream to 1e-7
 *      rather than bit for bit;
 *   5. consecutive overlap-save filters on a ComplexFloat32 stream merged into ONE filter with the taps convolved in double (up to 1 281
 *      taps): rounds once where the cascade rounds per stage, <= 1e-6 of the exact cascade.
 * lrhip_chain_create_ex() below switches 2-5 off per chain (LRHIP_CHAIN_EXACT); 1 is chosen per filter (use_fft = 0: direct form). */
lrhip_chain_t *lrhip_chain_create(lrhip_stage_t **stages, unsigned nstages);
/* The numerical contract per chain (what a LuaRadio script sets as DeviceChainBlock.exact, lua/radio/composites/devicechain.lua).
 * lrhip_chain_create(stages, n) == lrhip_chain_create_ex(stages, n, 0).  Flags:
 *   LRHIP_CHAIN_EXACT_ROTATOR      a frequency translator fused in front of a filter keeps the stand-alone translator's phasors
 *                                  (block-of-8 staging): fused == unfused and chunked == unchunked bit for bit, at 0.19 instead of
 *                                  0.16 ms per 2^26 samples of the WBFM receiver;
 *   LRHIP_CHAIN_NO_POLYPHASE_TAIL  FIR -> single-pole IIR -> downsampler keeps the blocks' own arithmetic (filter at the high rate,
 *                                  recurrence, gather) instead of ONE decimating filter with the recurrence at the low rate; and
 *                                  consecutive overlap-save filters on a ComplexFloat32 stream stay separate filters instead of ONE
 *                                  filter with the convolved taps (up to 1 281 taps, one 4096-point launch);
 *   LRHIP_CHAIN_NO_FUSION          every stage runs its own kernels (edges still device-resident);
 *   LRHIP_CHAIN_NO_SINGLE_LAUNCH   the FM receivers keep tuner + discriminator and audio tail as two launches (the round-2 form).
 * LRHIP_CHAIN_EXACT = EXACT_ROTATOR | NO_POLYPHASE_TAIL | NO_SINGLE_LAUNCH: the chain computes what its blocks compute one by one (bit for bit
 * for direct-form filters; overlap-save filters stay within 1e-6, as in the reference).  The environment variables LRHIP_TUNER_EXACT,
 * LRHIP_NO_POLYPHASE_TAIL, ... of DESIGN.md 4.5 remain as process-wide overrides for A/B measurements.  Unknown bits -> NULL. */
enum {
    LRHIP_CHAIN_EXACT_ROTATOR = 1,
    LRHIP_CHAIN_NO_POLYPHASE_TAIL = 2,
    LRHIP_CHAIN_NO_FUSION = 4,
    LRHIP_CHAIN_NO_SINGLE_LAUNCH = 8,
    LRHIP_CHAIN_EXACT = 1 | 2 | 8
};
lrhip_chain_t *lrhip_chain_create_ex(lrhip_stage_t **stages, unsigned nstages, unsigned flags);
void lrhip_chain_destroy(lrhip_chain_t *c);
/* Back to the initial state (zero history, phase, indices) for every stage of the chain, including the fused ones it built. */
int   lrhip_chain_reset(lrhip_chain_t *c);
unsigned long lrhip_chain_max_output(const lrhip_chain_t *c, unsigned long n_in);
long lrhip_chain_execute(lrhip_chain_t *c, const void *in_host, unsigned long n_in,
                         void *out_host, unsigned long out_capacity);
long lrhip_chain_execute_device(lrhip_chain_t *c, const void *in_dev, unsigned long n_in,
                                void *out_dev, unsigned long out_capacity);
/* Pipelined host path: the chain owns a RING of `depth` slots, each a pinned host input/output buffer plus device
 * input/output buffers sized for `max_chunk` input samples.  submit() copies the caller's vector into the next
 * pinned slot and enqueues H2D (copy-in stream) -> the chain's kernels (compute stream) -> D2H (copy-out stream),
 * chained by events, and returns at once; collect() waits for the OLDEST submitted chunk and hands back its output.
 * With depth >= 2 the PCIe transfers of neighbouring chunks overlap the kernels, and the block's process() can return
 * chunk k-depth+1 while chunk k is in flight (the reference's own FFT FIR also delays its output, firfilter.lua:361-398).
 * Block state is advanced at submit time, so results are identical to lrhip_chain_execute() chunk for chunk.
 * submit() returns the number of output samples the chunk WILL produce (>= 0) or < 0; collect() returns the number of
 * samples written, or < 0 (-2: nothing in flight).  This is what radio/core/pipe.lua's read/write loop
 * (pipe.lua:495-533, :252-262) becomes for a device chain. */
int  lrhip_chain_set_ring(lrhip_chain_t *c, unsigned depth, unsigned long max_chunk);
long lrhip_chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in);
/* Zero-copy input: the pinned host buffer of the slot the NEXT lrhip_chain_submit() will use (max_chunk input samples), or
 * NULL when the ring is full.  A source can read()/recv() straight into it and pass the same pointer to
 * lrhip_chain_submit(), which then skips its staging copy. */
void *lrhip_chain_ring_input(lrhip_chain_t *c);
/* File-fed chains: what IQFileSource / RealFileSource do per chunk (radio/blocks/sources/iqfile.lua:82-96: fread of raw records) without the interpreter or a
 * staging copy in the data path.  Reads up to max_in samples - raw records of lrhip_stage_input_size(first stage) bytes, max_in clamped to the ring's max_chunk -
 * of the REGULAR file `fd` from byte offset `offset` (positional reads: the descriptor's own position is neither used nor moved) straight into the pinned input of
 * the next ring slot, split over the library's copy threads (one read(2) stream gets ~10 GB/s out of the page cache, several get several times that), and submits
 * the slot as lrhip_chain_submit() does.  Returns the number of samples read and submitted; 0 at the end of the file (nothing submitted; a trailing partial
 * record is ignored, as fread() ignores it); < 0 on error: -3 ring full (collect first), -4 `fd` is not a regular file (pipes, sockets, devices: read() into
 * lrhip_chain_ring_input() and call lrhip_chain_submit()). */
long lrhip_chain_submit_fd(lrhip_chain_t *c, int fd, unsigned long long offset, unsigned long max_in);
long lrhip_chain_collect(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
/* Chunks submitted and not yet collected. */
int  lrhip_chain_in_flight(const lrhip_chain_t *c);
/* Chunk coalescing on top of the ring (what a DeviceChainBlock's process() calls): LuaRadio hands blocks whatever a read() returned,
 * 8 192 samples from a file source (radio/blocks/sources/iqfile.lua:52), up to 131 072 from a pipe (radio/core/pipe.lua:495-533) -
 * far too little for one launch set.  push() appends the caller's vector to the pinned input of the current ring slot and returns at
 * once; a slot is launched (H2D, kernels, D2H, asynchronously as for submit()) when it holds max_chunk samples, and the outputs of
 * slots that have finished are copied to out_host in stream order.  Sample VALUES are those of lrhip_chain_execute() on the same
 * stream (block state does not depend on chunking); only the emission is delayed, as the reference's own FFT filter delays its
 * output (firfilter.lua:361-398).  Returns the number of output samples written (>= 0, possibly 0) or < 0.
 * out_capacity must be at least lrhip_chain_push_bound(c, n_in).
 * lrhip_chain_flush(): launch the partly filled slot, wait for everything in flight and return the remaining outputs
 * (out_capacity >= lrhip_chain_push_bound(c, 0)); the chain can be pushed to again afterwards.  Call at EOF / cleanup(). */
long lrhip_chain_push(lrhip_chain_t *c, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity);
long lrhip_chain_flush(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
unsigned long lrhip_chain_push_bound(const lrhip_chain_t *c, unsigned long n_in);
/* Latency bound for LIVE flow graphs (an SDR source at 1.1 MS/s needs ~1 s to fill a 2^20-sample batch; an audio-rate chain minutes):
 * when the oldest sample pushed and not yet launched has waited max_seconds (wall clock, checked at every push), push() launches the
 * partial batch and returns its output from the same call.  0 (the default) = batches run only when full (file / benchmark sources,
 * which deliver faster than real time).  Sample values do not depend on where batches are cut. */
int  lrhip_chain_set_latency(lrhip_chain_t *c, double max_seconds);
/* The wall-clock side of that bound, for a source that STALLS: push() can only look at the clock when it is called, so a live graph whose
 * upstream goes quiet would leave the partial batch unlaunched, where the reference streams every chunk through as it arrives
 * (radio/core/block.lua:575-602).  The host waits for input at most lrhip_chain_poll_due() seconds (-1: nothing pending or no latency bound -
 * wait forever, as PipeMux:_read_single does, radio/core/pipe.lua:495-533; 0: due now) and calls lrhip_chain_poll() when that wait timed
 * out: a partial batch whose oldest sample has waited max_seconds is launched and its output returned from the call, finished batches are
 * handed out either way, and with nothing due it returns 0 at once.  out_capacity >= lrhip_chain_push_bound(c, 0). */
long lrhip_chain_poll(lrhip_chain_t *c, void *out_host, unsigned long out_capacity);
double lrhip_chain_poll_due(const lrhip_chain_t *c);
/* Number of kernels launched by the last chain execute (diagnostic for the fusion tests). */
int lrhip_chain_last_launches(const lrhip_chain_t *c);

/* ---- time-axis sharding: G partitions of ONE stream, each on its own GPU (or as G virtual partitions on one) -------------------------
 * The reference runs a block as one process over the whole stream (radio/core/block.lua:572-590); on a node with 8 GPUs a long
 * recording can be cut into G contiguous partitions instead, because every piece of cross-chunk state of the hot path is either a
 * closed-form function of the ABSOLUTE sample index (rotator phase = omega * n, frequencytranslator.lua:93-110; downsampler phase
 * = n mod D, downsampler.lua:45-56) or a bounded memory of the input (filter history M-1 samples, firfilter.lua:244-250;
 * the discriminator's previous sample, frequencydiscriminator.lua:74; a decaying recurrence's state, iirfilter.lua:113-181).
 *   lrhip_stage_seek / lrhip_chain_seek(c, n0): forget all carried samples and set the absolute counters as if n0 input samples
 *       had been consumed.  The next execute() is then sample n0 of the stream.
 *   lrhip_chain_halo(c): the number of input samples H a partition has to replay in front of its first own sample - seek(n0 - H),
 *       execute(x[n0-H .. n0)) with the output discarded - so that every carried state equals the uninterrupted stream's (filter
 *       histories exactly; recurrences to Float32 underflow of their zero start).  -1 (with lrhip_strerror) when a stage of the
 *       chain has unbounded memory (AGC, frequency modulator, a recurrence that does not decay): such a chain cannot be sharded
 *       in time.  Outputs of the partition [n0, n1) are then the same VALUES as samples of the single-stream run (to the Float32
 *       rounding stated at lrhip_chain_create: <= 1e-7 for the default single-launch receiver and overlap-save filters); with
 *       partition boundaries on multiples of lrhip_chain_shard_align(c) input samples AND a chain built with LRHIP_CHAIN_NO_SINGLE_LAUNCH
 *       (or LRHIP_CHAIN_EXACT) on direct-form filters also bit for bit.
 *   lrhip_chain_shard_align(c): partition boundaries on multiples of this many input samples give every scan kernel of the chain the
 *       tile grid of the uninterrupted run (1 for chains of filters / rotators / discriminators / downsamplers; 5 120 for a tuner
 *       fused with the discriminator behind it, whose tiles are rotated relative to their first sample; 128 000 for the WBFM receiver:
 *       the lcm of that and of the 2 560 audio samples x 25 of its tail). */
int  lrhip_stage_seek(lrhip_stage_t *q, unsigned long long n0);
int  lrhip_chain_seek(lrhip_chain_t *c, unsigned long long n0);
long lrhip_chain_halo(const lrhip_chain_t *c);
unsigned long lrhip_chain_shard_align(const lrhip_chain_t *c);
/* All of the above for a partition that owns the stream from `first_sample` on: seeks to the aligned sample s <= first_sample - halo
 * (returned in *seek_sample: the source has to deliver the stream from s on) and arms the chain to DROP the output of the first
 * (first_sample - s) input samples it is given - by execute, submit or push alike - so the caller's first output sample is the one the
 * single-stream run produces for input sample first_sample.  < 0 for chains with unbounded memory. */
int  lrhip_chain_start_at(lrhip_chain_t *c, unsigned long long first_sample, unsigned long long *seek_sample);

/* ---- fan-out across processes / GPUs below the host language (radio/core/pipe.lua:617-627: one output port, several readers) -------
 * LuaRadio runs every block in its own process; with one process per GPU the source's slab has to reach the other processes' devices
 * without a host round trip.  These are the HIP primitives for that, wrapped so that a LuaJIT host needs no HIP headers:
 *   a consumer process allocates its slab buffers (lrhip_malloc) and exports them (64-byte handles, sent over the control socket the
 *   host already has); the producer opens them and pushes each slab with lrhip_peer_copy on its COPY stream, then records an
 *   interprocess event the consumer waits on (on the GPU, lrhip_ipc_event_wait: no host synchronisation) before its chain reads the
 *   slab.  Two slabs per consumer double-buffer the copy of slab k+1 against the kernels on slab k; a second event per slab, recorded
 *   by the consumer after its chain and waited on by the producer, returns the buffer.  Over xGMI a receiving GPU is bounded by one
 *   link (about 153 GB/s = 19 GS/s of ComplexFloat32). */
enum { LRHIP_IPC_HANDLE_BYTES = 64 };          /* size of the opaque handles below */
int   lrhip_ipc_export(const void *dev_ptr, void *handle_out);                 /* hipIpcGetMemHandle */
void *lrhip_ipc_open(const void *handle);                                       /* hipIpcOpenMemHandle (lazy peer access) */
int   lrhip_ipc_close(void *dev_ptr);
typedef struct lrhip_ipc_event lrhip_ipc_event_t;
lrhip_ipc_event_t *lrhip_ipc_event_create(void *handle_out);                   /* interprocess event + its 64-byte handle */
lrhip_ipc_event_t *lrhip_ipc_event_open(const void *handle);
void  lrhip_ipc_event_destroy(lrhip_ipc_event_t *e);
int   lrhip_ipc_event_record(lrhip_ipc_event_t *e, int on_copy_stream);         /* on the copy stream (1) or the library stream (0) */
int   lrhip_ipc_event_wait(lrhip_ipc_event_t *e, int on_copy_stream);           /* the stream waits on the GPU; the host does not block */
int   lrhip_ipc_event_query(lrhip_ipc_event_t *e);                              /* 1 done, 0 pending, < 0 error */
int   lrhip_ipc_event_synchronize(lrhip_ipc_event_t *e);                        /* host wait */
/* dst/src may live on different devices (peer access is enabled on first use); asynchronous on the library's copy stream */
int   lrhip_peer_copy(void *dst, int dst_device, const void *src, int src_device, unsigned long bytes);
int   lrhip_copy_stream_synchronize(void);

/* ---- device memory helpers for FFI callers that want resident vectors ------------------------------------- */
void *lrhip_malloc(unsigned long bytes);
void  lrhip_free(void *dev_ptr);
int   lrhip_memcpy_h2d(void *dev_dst, const void *host_src, unsigned long bytes);
int   lrhip_memcpy_d2h(void *host_dst, const void *dev_src, unsigned long bytes);
/* device-to-device, asynchronous on the library stream (non-overlapping ranges) */
int   lrhip_memcpy_d2d(void *dev_dst, const void *dev_src, unsigned long bytes);
/* Pinned host vectors (what radio/core/vector.lua's platform.alloc would return for device-fed blocks). */
void *lrhip_host_alloc(unsigned long bytes);
void  lrhip_host_free(void *host_ptr);
/* Zero-copy for vectors the CALLER owns: LuaRadio's owning vectors are page-aligned and long-lived (posix_memalign, radio/core/vector.lua:19-37), its pipe
 * read buffers likewise (platform.alloc, radio/core/pipe.lua:72-76).  A registered range is pinned where it lies (hipHostRegister); the synchronous
 * host-pointer entry points (lrhip_stage_execute, lrhip_chain_execute) then DMA from / to it directly instead of staging through the library's own pinned
 * buffers - whenever the WHOLE input (or output) vector of a call lies inside one registered range.  The owner unregisters before it frees or reallocates
 * (a Vector that grew: radio/core/vector.lua:108-136).  Registrations do not survive fork().  Unregistered vectors work as before.
 * Round 5: with BOTH vectors of a call registered, a stage or chain whose first kernel reads its input once and whose last kernel writes its output once
 * (the FIR forms, Tuner / Decimator, the FM receiver) is handed the host vectors themselves - its loads and stores cross the link, one launch per call, no
 * staging on the device.  The values are those of the same call on device-resident vectors (an overlap-save filter sees the call as ONE chunk, where the
 * staged path cuts calls of 2^20 samples and more into pieces that are chunks of their own: Float32 rounding apart, include the statement above).
 * Round 6: the same for FrequencyTranslator, Downsampler and the complex -> real element-wise blocks (where it measured faster than the staged pipeline);
 * NOT for a filter created with the reference's block-emission framing (use_fft = 1: it copies its input first), and not when the input and output ranges
 * of a call overlap (an in-place call is served by the staged path, as one piece: the whole input is on the device before the first output byte returns). */
int   lrhip_host_register(void *host_ptr, unsigned long bytes);
int   lrhip_host_unregister(void *host_ptr);

/* ---- timing on the library stream (HIP events; used by bench.py for the roofline figure) ------------------- */
typedef struct lrhip_timer lrhip_timer_t;
lrhip_timer_t *lrhip_timer_create(void);
void  lrhip_timer_destroy(lrhip_timer_t *t);
int   lrhip_timer_start(lrhip_timer_t *t);   /* records an event on the current stream */
int   lrhip_timer_stop(lrhip_timer_t *t);    /* records the closing event */
double lrhip_timer_elapsed_ms(lrhip_timer_t *t);   /* waits for the closing event; < 0 on error */

#ifdef __cplusplus
}
#endif
#endif /* LRHIP_H */
