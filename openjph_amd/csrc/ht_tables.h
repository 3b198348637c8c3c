// openjph_amd/csrc/ht_tables.h -- HT block coder look-up tables, built on the host from the
// CxtVLC rows of ITU-T T.814 Annex C (ht_vlc_tables.inc) and uploaded once per device.
//
// Same table *contents* as the reference builds at start-up (encoder: vlc_init_tables,
// ojph_block_encoder.cpp:76-193; decoder: vlc_init_tables / uvlc_init_tables,
// ojph_block_common.cpp:124-336) -- the kernels index them the same way.
#ifndef OJPH_HT_TABLES_H
#define OJPH_HT_TABLES_H
#include <stdint.h>

namespace ojphgpu {

struct HtTables {
  uint16_t enc_vlc[2][2048];   // [(c_q << 8) | (rho << 4) | eps] -> (cwd << 8) | (len << 4) | e_k
  uint16_t dec_vlc[2][1024];   // [(c_q << 7) | 7 bits] -> e_k<<12 | e_1<<8 | rho<<4 | u_off<<3 | len
  // the fused launch's table: bits 0..7 as above; bits 8..10 what the chain needs next, derived from rho (ht_tables.cpp);
  // and in the upper half what step 2 needs of the entry in 9 bits (the fused launch's 16-bit records):
  // bits 16 + 2i, 17 + 2i: sample i of the quad -- 0 insignificant, 1 significant, 2 significant with its e_k bit, 3 with
  // e_k and e_1 (e_1 is a subset of e_k, e_k of rho in every row of the standard's tables); bit 24: more than one sample
  // significant (gamma of T.814, block_decoder32.cpp:1218)
  uint32_t dec_vlc32[2][1024];
  uint16_t dec_uvlc0[320];     // initial quad row
  uint16_t dec_uvlc1[256];     // other rows
};

void build_ht_tables(HtTables& t);
// uploads the tables to the current HIP device (once per device); 0 on success
int ensure_tables();
int upload_enc_tables(const HtTables& t);
int upload_dec_tables(const HtTables& t);

}  // namespace ojphgpu

struct ojphgpu_cb_desc; struct ojphgpu_cb_result;
namespace ojphgpu {
// ojphgpu_ht_encode with the caller's knowledge of which block widths the range holds (kernels_ht_enc.hip)
int ht_encode_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const void* d_coef, uint8_t* d_scratch,
                     uint8_t* d_out, uint32_t out_cap, ojphgpu_cb_result* d_results, uint32_t* d_cursor, uint32_t* d_status,
                     int widths, const uint32_t* d_regions, uint32_t nreg);   // regions: see claim_output (kernels_ht_enc.hip)
// ojphgpu_ht_decode_step2 with the caller's knowledge of the blocks of the range (kernels_ht_dec.hip)
// step 1 + step 2 in one launch (kernels_ht_dec.hip, ht_dec_fused_kernel): chains first, step-2 workers behind them
bool dec_fuses();
uint64_t ht_decode_fused_state_words(uint32_t n);
uint32_t device_cus(int device);                          // compute units of that device (256 when the query fails)
bool ht_decode_fused_pays(uint32_t n, uint32_t max_h, uint32_t cus);   // one launch for step 1 + step 2, or the separate launches
uint32_t ht_decode_fused_grid(uint32_t n, uint32_t cus);  // workgroups of that launch
int ht_decode_fused_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data, uint32_t* d_quad_scratch,
                           void* d_coef, uint8_t* d_block_status, uint32_t* d_state, uint32_t epoch, uint32_t max_h, int kinds,
                           uint32_t cus, uint32_t* d_host_retry);
// the blocks of a range that are on the 64-bit sample path (ojph_decode_codeblock64): all their launches
uint32_t ht_decode64_extra_aux_words(uint32_t len1);
int ht_decode64_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data, uint32_t* d_aux,
                       uint32_t* d_quad_scratch, void* d_coef, uint8_t* d_block_status, int refine);
int ht_decode_step2_launch(void* stream, const ojphgpu_cb_desc* d_blocks, uint32_t n, const uint8_t* d_data,
                           const uint32_t* d_quad_scratch, void* d_coef, uint8_t* d_block_status, int kinds);

}  // namespace ojphgpu
#endif
