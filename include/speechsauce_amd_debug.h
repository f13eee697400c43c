/*
 * speechsauce_amd_debug.h -- process-wide test aids of the LAB build, libspeechsauce_amd_lab.so (`make -C mfcc-rust_amd/csrc lab`,
 * the same sources with -DSS_LAB=1).
 *
 * NOT exported by the product library libspeechsauce_amd.so (round 4): its header promises an immutable, thread-safe handle and
 * a kernel selection that is a pure function of the configuration and the call, and switches that change kernel selection or
 * inject faults for every config in the process do not belong in what ships.  tests/ load the lab library explicitly where
 * they need one of these (tests/conftest.py `sslab`): LDS poisoning before product-library launches, forced kernel builds for
 * the bit-for-bit comparisons, the lost-hand-off fault of the retired tile build.  All settings are process-wide and meant for
 * single-threaded test drivers.  The shader clock bench.py reports comes from the product's own per-call diagnostic,
 * ss_mfcc_shader_clock (speechsauce_amd.h).
 */
#ifndef SPEECHSAUCE_AMD_DEBUG_H
#define SPEECHSAUCE_AMD_DEBUG_H

#include "speechsauce_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fills the LDS of every compute unit with 0xFFFFFFFF (a NaN pattern as f32, -1 as i32).  LDS is not cleared between
 * kernels, so a kernel that lets a word it never wrote reach its results fails loudly after this. */
int ss_debug_poison_lds(void *stream);

/* While `d_stamps` is non-null, every launch of the 512-point MFCC kernel writes six 64-bit words per wave into it
 * (16 waves x CUs x 6 words, overwritten by each launch): [0] s_memrealtime (100 MHz) at wave start, [2] at wave end,
 * [5] shader-clock cycles the wave lived | 1 << 40 (the two table waves of a workgroup: ticks from their start until the tables
 * were in LDS | 2 << 40), [1] / [3] / [4] prologue end,
 * quads done << 32 | XCC id, first samples arrived (tools/prof2.py, tools/dbg_times.py).  Pass NULL to switch it off. */
int ss_debug_stamp_buffer(unsigned long long *d_stamps);

/* on != 0: every configuration runs on the generic kernel (ss_front_generic) instead of its dedicated one -- the
 * cross-check of tests/test_gpu_parity.py::test_kernel_variants_agree and the "generic us" columns of DESIGN.md. */
int ss_debug_force_generic(int on);

/* Which build of the 2048-point mel-spectrogram kernel runs (A/B and the bit-for-bit comparison of the builds):
 * mode 1: automatic (default); 0: eight waves per CU with direct stores; 2: the eight-wave builds only -- the retired
 * whole-line tile (tools/experiments/ss_mel2048_tile.hip, lab library only) when the batch allows it; 3: the twelve-wave
 * build wherever it exists. */
int ss_debug_mel_tile(int mode);

/* on != 0: the next launches of ss_mel_c1024<tile> do not poll at all -- a wave whose tile hand-off (a clip's last row pair,
 * a buffer's release) is not there the moment it looks takes the lost-hand-off path: it sets the config's device error
 * word and ends.  The launch must end (no hang) and the config must report SS_ERR_DEVICE.  on == 0: normal operation
 * (2^24 polls, about half a second, before a hand-off counts as lost). */
int ss_debug_tile_fault(int on);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHSAUCE_AMD_DEBUG_H */
