/* oracle/ht_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's (aous72/OpenJPH 0.31.0) HTJ2K hot path: HT block
 * coder, 5/3 and 9/7 DWT, quantise transfer, colour / sample conversion.  It is the parity
 * checker for the HIP kernels in openjph_amd/csrc; the product never links, loads or calls it.
 * Pinned against the real reference (oracle/_ref) by tests/test_oracle_vs_ref.py.
 */
#ifndef HT_ORACLE_H
#define HT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* HT cleanup-pass encoder of one code-block (reference ojph_block_encoder.cpp:542-1017).
 * buf: sign-magnitude samples (bit31 sign, magnitude MSB-aligned), returns byte count (0 on
 * overflow of `cap`).  variant 0 = sequential writers, 1 = flat-bitstream + stuffing passes (the
 * formulation the HIP kernel uses); both must give identical bytes. */
int ojo_ht_encode(const uint32_t* buf, int width, int height, int stride, int missing_msbs,
                  uint8_t* out, int cap, int variant);

/* HT decoder (cleanup [+ SigProp + MagRef]) of one code-block (ojph_block_decoder32.cpp:742).
 * returns 1 on success, 0 on failure. */
int ojo_ht_decode(const uint8_t* coded, int len1, int len2, int num_passes, int missing_msbs,
                  int width, int height, int stride, uint32_t* out, int stripe_causal);

/* The same two functions on 64-bit sign-magnitude samples (sign in bit 63): ojph_encode_codeblock64
 * (ojph_block_encoder.cpp:1026) / ojph_decode_codeblock64 (ojph_block_decoder64.cpp:766) -- the sample path the
 * reference takes when a component needs more than 32 bits of precision (ojph_params.cpp:1684-1706). */
int ojo_ht_encode64(const uint64_t* buf, int width, int height, int stride, int missing_msbs,
                    uint8_t* out, int cap, int variant);
int ojo_ht_decode64(const uint8_t* coded, int len1, int len2, int num_passes, int missing_msbs,
                    int width, int height, int stride, uint64_t* out, int stripe_causal);

/* one DWT level over a whole plane.  src: w x h (pitch sp).  Sub-band planes: LL is
 * lw x lh, HL is hw x lh, LH is lw x hh, HH is hw x hh with lw = (w + x_even) >> 1,
 * hw = (w + !x_even) >> 1 (same for rows).  x_even / y_even: parity of the plane origin
 * (1 = first sample at an even canvas coordinate).  Analysis is vertical then horizontal,
 * synthesis horizontal then vertical (ojph_resolution.cpp:547-949, ojph_transform.cpp:209-852). */
void ojo_dwt53_fwd(const int32_t* src, int sp, int w, int h, int x_even, int y_even,
                   int32_t* ll, int llp, int32_t* hl, int hlp, int32_t* lh, int lhp,
                   int32_t* hh, int hhp);
void ojo_dwt53_inv(int32_t* dst, int dp, int w, int h, int x_even, int y_even,
                   const int32_t* ll, int llp, const int32_t* hl, int hlp, const int32_t* lh,
                   int lhp, const int32_t* hh, int hhp);
void ojo_dwt97_fwd(const float* src, int sp, int w, int h, int x_even, int y_even,
                   float* ll, int llp, float* hl, int hlp, float* lh, int lhp, float* hh, int hhp);
void ojo_dwt97_inv(float* dst, int dp, int w, int h, int x_even, int y_even,
                   const float* ll, int llp, const float* hl, int hlp, const float* lh, int lhp,
                   const float* hh, int hhp);

/* One DWT level in its general form: lifting steps of an ATK marker segment (synthesis order; reversible steps use
 * a, b, e, irreversible ones A and the scaling K), one direction only where a DFS marker segment says so (horz / vert),
 * elem 0 = int32, 1 = int64, 2 = float.  Plane and band geometry as above; with vert = 0 every row is a "low" row (ll
 * and hl are produced, lh / hh unused), with horz = 0 every column a "low" column (ll and lh).
 * (ojph_transform.cpp:209-852, ojph_resolution.cpp:547-949, ojph_params.cpp:2654-2896) */
typedef struct { int32_t a, b, e; float A; } ojo_lift_step;
void ojo_dwt_fwd_gen(const void* src, int sp, int w, int h, int x_even, int y_even, int elem, int horz, int vert,
                     const ojo_lift_step* steps, int nsteps, float K,
                     void* ll, int llp, void* hl, int hlp, void* lh, int lhp, void* hh, int hhp);
void ojo_dwt_inv_gen(void* dst, int dp, int w, int h, int x_even, int y_even, int elem, int horz, int vert,
                     const ojo_lift_step* steps, int nsteps, float K,
                     const void* ll, int llp, const void* hl, int hlp, const void* lh, int lhp, const void* hh, int hhp);

/* quantise transfer (ojph_codestream_gen.cpp:59-181). Return OR of magnitudes. */
uint64_t ojo_quant_rev64(const int64_t* src, uint64_t* dst, int count, int K_max);
void ojo_dequant_rev64(const uint64_t* src, int64_t* dst, int count, int K_max);
void ojo_rev_convert_to64(const int32_t* src, int64_t* dst, int count, int64_t shift, int nlt3);
void ojo_rev_convert_from64(const int64_t* src, int32_t* dst, int count, int64_t shift, int nlt3);
void ojo_rct_fwd64(const int32_t* r, const int32_t* g, const int32_t* b, int64_t* y, int64_t* cb, int64_t* cr, int count);
void ojo_rct_inv64(const int64_t* y, const int64_t* cb, const int64_t* cr, int32_t* r, int32_t* g, int32_t* b, int count);
uint32_t ojo_quant_rev(const int32_t* src, uint32_t* dst, int count, int K_max);
uint32_t ojo_quant_irv(const float* src, uint32_t* dst, int count, float delta_inv);
void ojo_dequant_rev(const uint32_t* src, int32_t* dst, int count, int K_max);
void ojo_dequant_irv(const uint32_t* src, float* dst, int count, float delta);

/* sample conversion + colour transforms (ojph_colour.cpp:238-571). */
void ojo_rev_convert(const int32_t* src, int32_t* dst, int count, int32_t shift);
void ojo_irv_to_float(const int32_t* src, float* dst, int count, int bit_depth, int is_signed);
void ojo_irv_to_int(const float* src, int32_t* dst, int count, int bit_depth, int is_signed);
void ojo_rct_fwd(const int32_t* r, const int32_t* g, const int32_t* b, int32_t* y, int32_t* cb,
                 int32_t* cr, int count);
void ojo_rct_inv(const int32_t* y, const int32_t* cb, const int32_t* cr, int32_t* r, int32_t* g,
                 int32_t* b, int count);
void ojo_ict_fwd(const float* r, const float* g, const float* b, float* y, float* cb, float* cr,
                 int count);
void ojo_ict_inv(const float* y, const float* cb, const float* cr, float* r, float* g, float* b,
                 int count);

#ifdef __cplusplus
}
#endif
#endif
