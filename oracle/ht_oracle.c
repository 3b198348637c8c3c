/* oracle/ht_oracle.c -- TEST INFRASTRUCTURE ONLY (see ht_oracle.h).
 *
 * Plain-C CPU restatement of the HTJ2K hot path of aous72/OpenJPH 0.31.0.  Each function cites
 * the reference file:line it follows.  The code is written from the behaviour of the reference
 * (and of ITU-T T.814), not copied: sample neighbourhoods are computed directly from the
 * block instead of through rolling line state, bit-streams are modelled as flat bit arrays, and
 * the quad pair -- not the image line -- is the unit of work, because that is the shape the
 * one-wavefront-per-code-block HIP kernels use.
 *
 * Parity status: PINNED.  tests/test_cpu_parity.py checks every function here against the real
 * reference compiled from /root/reference (oracle/_ref/libojph_ref*.so, oracle/Makefile) and against
 * the golden vectors made from it (tests/golden/); tests/test_survey_ka.py pins the survey's
 * whole-image known answers KA-3 / KA-4.
 */
#include "ht_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "ht_vlc_tables.inc"

/* ------------------------------------------------------------------------------------------ */
/* tables                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static uint16_t enc_vlc[2][2048];  /* (cwd << 8) | (len << 4) | e_k     : block_encoder.cpp:61-193 */
static uint16_t dec_vlc[2][1024];  /* e_k<<12|e_1<<8|rho<<4|u_off<<3|len : block_common.cpp:124-190 */
static uint16_t dec_uvlc0[320];    /* block_common.cpp:196-292 */
static uint16_t dec_uvlc1[256];    /* block_common.cpp:294-336 */
static int tables_ready = 0;

#define ROW_CQ(r)  ((int)((r) & 7))
#define ROW_RHO(r) ((int)(((r) >> 3) & 15))
#define ROW_UOFF(r) ((int)(((r) >> 7) & 1))
#define ROW_EK(r)  ((int)(((r) >> 8) & 15))
#define ROW_E1(r)  ((int)(((r) >> 12) & 15))
#define ROW_CWD(r) ((int)(((r) >> 16) & 127))
#define ROW_LEN(r) ((int)(((r) >> 23) & 7))

static int popcnt4(int v) { return (v & 1) + ((v >> 1) & 1) + ((v >> 2) & 1) + ((v >> 3) & 1); }

static void build_tables(void)
{
  if (tables_ready) return;
  const unsigned int* src[2] = { HT_VLC_SRC0, HT_VLC_SRC1 };
  int n[2] = { (int)(sizeof(HT_VLC_SRC0) / 4), (int)(sizeof(HT_VLC_SRC1) / 4) };
  for (int t = 0; t < 2; ++t) {
    /* encoder look-up: index (c_q << 8) | (rho << 4) | eps   (block_encoder.cpp:91-135) */
    for (int i = 0; i < 2048; ++i) {
      int c_q = i >> 8, rho = (i >> 4) & 15, emb = i & 15;
      enc_vlc[t][i] = 0;
      if ((emb & rho) != emb || (rho == 0 && c_q == 0)) continue;
      int best = -1;
      if (emb) {
        int best_cnt = -1;
        for (int j = 0; j < n[t]; ++j) {
          unsigned r = src[t][j];
          if (ROW_CQ(r) == c_q && ROW_RHO(r) == rho && ROW_UOFF(r) == 1 &&
              (emb & ROW_EK(r)) == ROW_E1(r)) {
            int cnt = popcnt4(ROW_EK(r));
            if (cnt >= best_cnt) { best = j; best_cnt = cnt; }
          }
        }
      } else {
        for (int j = 0; j < n[t]; ++j) {
          unsigned r = src[t][j];
          if (ROW_CQ(r) == c_q && ROW_RHO(r) == rho && ROW_UOFF(r) == 0) { best = j; break; }
        }
      }
      if (best >= 0) {
        unsigned r = src[t][best];
        enc_vlc[t][i] = (uint16_t)((ROW_CWD(r) << 8) | (ROW_LEN(r) << 4) | ROW_EK(r));
      }
    }
    /* decoder look-up: index (c_q << 7) | 7 stream bits      (block_common.cpp:155-187) */
    for (int i = 0; i < 1024; ++i) {
      int cwd = i & 0x7F, c_q = i >> 7;
      dec_vlc[t][i] = 0;
      for (int j = 0; j < n[t]; ++j) {
        unsigned r = src[t][j];
        if (ROW_CQ(r) == c_q && ROW_CWD(r) == (cwd & ((1 << ROW_LEN(r)) - 1)))
          dec_vlc[t][i] = (uint16_t)((ROW_RHO(r) << 4) | (ROW_UOFF(r) << 3) | (ROW_EK(r) << 12) |
                                     (ROW_E1(r) << 8) | ROW_LEN(r));
      }
    }
  }
  /* UVLC prefix decode: index = 3 LSBs of the stream; value = prefix_len | suffix_len<<2 |
   * u_pfx<<5 (T.814 table 3; block_common.cpp:204-213) */
  static const uint8_t pfx[8] = {
    3 | (5 << 2) | (5 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5),
    3 | (1 << 2) | (3 << 5), 1 | (0 << 2) | (1 << 5), 2 | (0 << 2) | (2 << 5), 1 | (0 << 2) | (1 << 5) };
  for (int i = 0; i < 320; ++i) {
    int mode = i >> 6, vlc = i & 0x3F;
    unsigned tp = 0, ts = 0, s0 = 0, u0 = 0, u1 = 0;
    if (mode == 1 || mode == 2) {
      unsigned d = pfx[vlc & 7];
      tp = d & 3; ts = (d >> 2) & 7;
      s0 = (mode == 1) ? ts : 0;
      u0 = (mode == 1) ? (d >> 5) : 0;
      u1 = (mode == 1) ? 0 : (d >> 5);
    } else if (mode == 3) {
      unsigned d0 = pfx[vlc & 7];
      unsigned d1 = pfx[(vlc >> (d0 & 3)) & 7];
      if ((d0 & 3) == 3) { /* u_q0 > 2: second quad is signalled with a single bit */
        tp = (d0 & 3) + 1; s0 = (d0 >> 2) & 7; ts = s0; u0 = d0 >> 5;
        u1 = ((vlc >> (d0 & 3)) & 1) + 1;
      } else {
        tp = (d0 & 3) + (d1 & 3); s0 = (d0 >> 2) & 7; ts = s0 + ((d1 >> 2) & 7);
        u0 = d0 >> 5; u1 = d1 >> 5;
      }
    } else if (mode == 4) {
      unsigned d0 = pfx[vlc & 7];
      unsigned d1 = pfx[(vlc >> (d0 & 3)) & 7];
      tp = (d0 & 3) + (d1 & 3); s0 = (d0 >> 2) & 7; ts = s0 + ((d1 >> 2) & 7);
      u0 = (d0 >> 5) + 2; u1 = (d1 >> 5) + 2;
    }
    dec_uvlc0[i] = (uint16_t)(tp | (ts << 3) | (s0 << 7) | (u0 << 10) | (u1 << 13));
    if (i < 256) {
      if (mode == 3) { /* non-initial rows: no special cases */
        unsigned d0 = pfx[vlc & 7];
        unsigned d1 = pfx[(vlc >> (d0 & 3)) & 7];
        tp = (d0 & 3) + (d1 & 3); s0 = (d0 >> 2) & 7; ts = s0 + ((d1 >> 2) & 7);
        u0 = d0 >> 5; u1 = d1 >> 5;
      }
      dec_uvlc1[i] = (uint16_t)(tp | (ts << 3) | (s0 << 7) | (u0 << 10) | (u1 << 13));
    }
  }
  tables_ready = 1;
}

static inline int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline int clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

/* UVLC encoder code of u -> prefix / suffix / extension (block_encoder.cpp:196-255): u <= 32 has no extension (all the
 * 32-bit sample path ever produces); u >= 33 -- only the 64-bit sample path, :1269-1286 -- is coded as suffix 28 + (u - 33) % 4
 * and a 4-bit extension (u - 33) / 4 */
static void uvlc_code(int u, int* pre, int* pre_len, int* suf, int* suf_len, int* ext, int* ext_len)
{
  *ext = 0; *ext_len = 0;
  if (u == 0) { *pre = 0; *pre_len = 0; *suf = 0; *suf_len = 0; }
  else if (u == 1) { *pre = 1; *pre_len = 1; *suf = 0; *suf_len = 0; }
  else if (u == 2) { *pre = 2; *pre_len = 2; *suf = 0; *suf_len = 0; }
  else if (u <= 4) { *pre = 4; *pre_len = 3; *suf = u - 3; *suf_len = 1; }
  else if (u <= 32) { *pre = 0; *pre_len = 3; *suf = u - 5; *suf_len = 5; }
  else { *pre = 0; *pre_len = 3; *suf = 28 + (u - 33) % 4; *suf_len = 5; *ext = (u - 33) / 4; *ext_len = 4; }
}

/* ------------------------------------------------------------------------------------------ */
/* flat LSB-first bit array (the model of the per-wave LDS bit buffers)                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t* b; long nbits; long cap_bytes; } flatbits;

static void fb_init(flatbits* f, long cap_bytes)
{ f->b = (uint8_t*)calloc((size_t)cap_bytes + 8, 1); f->nbits = 0; f->cap_bytes = cap_bytes; }
static void fb_free(flatbits* f) { free(f->b); }
static int fb_put(flatbits* f, uint32_t v, int n)
{
  if (n <= 0) return 1;
  if ((f->nbits + n + 7) / 8 > f->cap_bytes) return 0;
  uint64_t w, m = (n >= 32) ? 0xFFFFFFFFull : ((1ull << n) - 1);
  long byte = f->nbits >> 3; int sh = (int)(f->nbits & 7);
  memcpy(&w, f->b + byte, 8);
  w |= ((uint64_t)v & m) << sh;
  memcpy(f->b + byte, &w, 8);
  f->nbits += n;
  return 1;
}
static int fb_put64(flatbits* f, uint64_t v, int n)        /* n up to 64 */
{
  if (n > 32) { if (!fb_put(f, (uint32_t)v, 32)) return 0; return fb_put(f, (uint32_t)(v >> 32), n - 32); }
  return fb_put(f, (uint32_t)v, n);
}
/* The reference's readers OR every raw byte into their window with all 8 bits and only then
 * advance by 7 or 8 (frwd_read :609-655, rev_read :308-359): after a stuffing event the byte's MSB
 * lands on the next byte's LSB.  In a conforming stream that MSB is 0; for arbitrary (corrupt)
 * bytes this is what keeps the restatement bit-identical. */
static int fb_put_byte(flatbits* f, uint32_t byte, int nb)
{
  if ((f->nbits + 8 + 7) / 8 > f->cap_bytes) return 0;
  uint64_t w; long at = f->nbits >> 3; int sh = (int)(f->nbits & 7);
  memcpy(&w, f->b + at, 8);
  w |= (uint64_t)(byte & 0xFFu) << sh;
  memcpy(f->b + at, &w, 8);
  f->nbits += nb;
  return 1;
}

static uint32_t fb_get(const flatbits* f, long pos, int n)
{
  if (n <= 0 || (pos >> 3) >= f->cap_bytes) return 0;
  uint64_t w; memcpy(&w, f->b + (pos >> 3), 8);
  w >>= (pos & 7);
  if (n < 32) w &= (1ull << n) - 1;
  return (uint32_t)w;
}
static uint64_t fb_get64(const flatbits* f, long pos, int n)   /* n up to 64 */
{
  uint64_t lo = fb_get(f, pos, n < 32 ? n : 32);
  if (n > 32) lo |= (uint64_t)fb_get(f, pos + 32, n - 32) << 32;
  return lo;
}

/* ------------------------------------------------------------------------------------------ */
/* sequential writers (variant 0)   block_encoder.cpp:273-534                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint8_t* buf; int pos, size; int rem, tmp, run, k, thr; } melw;
typedef struct { uint8_t* last; int pos, size; int used, tmp, gt8f; } vlcw;
typedef struct { uint8_t* buf; int pos, size; int maxb, used; uint32_t tmp; } msw;
static const int MEL_E[13] = { 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5 };

static int mel_bit(melw* m, int v)
{
  m->tmp = (m->tmp << 1) + v;
  if (--m->rem == 0) {
    if (m->pos >= m->size) return 0;
    m->buf[m->pos++] = (uint8_t)m->tmp;
    m->rem = (m->tmp == 0xFF) ? 7 : 8;
    m->tmp = 0;
  }
  return 1;
}
static int mel_event(melw* m, int bit)
{
  int ok = 1;
  if (!bit) {
    if (++m->run >= m->thr) {
      ok &= mel_bit(m, 1);
      m->run = 0; m->k = m->k < 12 ? m->k + 1 : 12; m->thr = 1 << MEL_E[m->k];
    }
  } else {
    ok &= mel_bit(m, 0);
    for (int t = MEL_E[m->k]; t > 0; ) ok &= mel_bit(m, (m->run >> --t) & 1);
    m->run = 0; m->k = m->k > 0 ? m->k - 1 : 0; m->thr = 1 << MEL_E[m->k];
  }
  return ok;
}
static int vlc_put(vlcw* v, int cwd, int len)
{
  while (len > 0) {
    if (v->pos >= v->size) return 0;
    int avail = 8 - v->gt8f - v->used;
    int t = avail < len ? avail : len;
    v->tmp |= (cwd & ((1 << t) - 1)) << v->used;
    v->used += t; avail -= t; len -= t; cwd >>= t;
    if (avail == 0) {
      if (v->gt8f && v->tmp != 0x7F) { v->gt8f = 0; continue; }
      *(v->last - v->pos) = (uint8_t)v->tmp; v->pos++;
      v->gt8f = v->tmp > 0x8F; v->tmp = 0; v->used = 0;
    }
  }
  return 1;
}
static int ms_put(msw* m, uint64_t cwd, int len)          /* ms_encode / ms_encode64 (:471-512) */
{
  while (len > 0) {
    if (m->pos >= m->size) return 0;
    int t = m->maxb - m->used; if (t > len) t = len;
    m->tmp |= (uint32_t)(cwd & ((1ull << t) - 1)) << m->used;
    m->used += t; cwd >>= t; len -= t;
    if (m->used >= m->maxb) {
      m->buf[m->pos++] = (uint8_t)m->tmp;
      m->maxb = (m->tmp == 0xFF) ? 7 : 8; m->tmp = 0; m->used = 0;
    }
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* HT cleanup encoder                                                                          */
/* ------------------------------------------------------------------------------------------ */
/* The coder is written once, on 64-bit sign-magnitude samples (sign in bit 63, magnitude MSB aligned below it), which is
 * ojph_encode_codeblock64 (block_encoder.cpp:1026-1520).  ojph_encode_codeblock32 (:542-1017) is the same algorithm on
 * 32-bit words: with p32 = 30 - missing_msbs and p64 = 62 - missing_msbs = p32 + 32, a 32-bit word moved up by 32 bits
 * gives the same 2*mu_p, hence the same symbols; it never reaches u > 32, so the U-VLC extension is never emitted. */
typedef struct { uint64_t val; uint32_t sign; } smp;

static inline smp get_smp(const uint64_t* buf, int w, int h, int stride, int x, int y, int p)
{
  smp s = { 0, 0 };
  if (x < 0 || y < 0 || x >= w || y >= h) return s;
  uint64_t t = buf[(size_t)y * stride + x];
  s.val = ((t + t) >> p) & ~1ull;   /* 2*mu_p : block_encoder.cpp:592-595, :1093-1096 */
  s.sign = (uint32_t)(t >> 63);
  return s;
}
static inline int expo(uint64_t val) { return val ? 64 - clz64(val - 1) : 0; }

typedef struct {         /* everything one quad contributes, in stream order */
  int exists, rho, c_q, u, U, tuple;
  uint64_t ms_val[4]; int ms_len[4];
  int mel_valid, mel_bit;
} quadsym;

static void quad_symbols(const uint64_t* buf, int w, int h, int stride, int p, int qx, int qy,
                         int rho_left, quadsym* q)
{
  memset(q, 0, sizeof(*q));
  if (2 * qx >= w) return;
  q->exists = 1;
  int e[4], emax = 0; uint64_t s[4];
  for (int n = 0; n < 4; ++n) {
    smp a = get_smp(buf, w, h, stride, 2 * qx + (n >> 1), 2 * qy + (n & 1), p);
    e[n] = expo(a.val);
    s[n] = a.val ? a.val - 2 + a.sign : 0;   /* v_n = 2(mu-1) + sign : block_encoder.cpp:601 */
    if (a.val) q->rho |= 1 << n;
    if (e[n] > emax) emax = e[n];
  }
  int kappa = 1, tbl = 0;
  if (qy == 0) {
    q->c_q = (rho_left >> 1) | (rho_left & 1);                 /* block_encoder.cpp:731,788 */
  } else {
    tbl = 1;
    int E[4], S[4];
    for (int i = 0; i < 4; ++i) {  /* row above, columns 2qx-1 .. 2qx+2 */
      smp a = get_smp(buf, w, h, stride, 2 * qx - 1 + i, 2 * qy - 1, p);
      E[i] = expo(a.val); S[i] = a.val != 0;
    }
    int max_e = E[0]; for (int i = 1; i < 4; ++i) if (E[i] > max_e) max_e = E[i];
    max_e -= 1;
    if (q->rho & (q->rho - 1)) kappa = max_e > 1 ? max_e : 1;   /* block_encoder.cpp:862 */
    q->c_q = (S[0] | S[1]) | ((S[2] | S[3]) << 2)               /* :802,:878,:967 */
           | ((rho_left & 4) >> 1) | ((rho_left & 8) >> 2);     /* :951,:991 */
  }
  q->U = emax > kappa ? emax : kappa;
  q->u = q->U - kappa;
  int eps = 0;
  if (q->u > 0) for (int n = 0; n < 4; ++n) eps |= (e[n] == emax) << n;
  q->tuple = enc_vlc[tbl][(q->c_q << 8) + (q->rho << 4) + eps];
  if (q->c_q == 0) { q->mel_valid = 1; q->mel_bit = q->rho != 0; }
  for (int n = 0; n < 4; ++n) {
    int m = (q->rho >> n) & 1 ? q->U - ((q->tuple >> n) & 1) : 0;   /* :667-674 */
    q->ms_len[n] = m;
    q->ms_val[n] = m ? (s[n] & ((m < 64 ? (1ull << m) : 0ull) - 1ull)) : 0;
  }
}

/* collects the (cwd,len) items of a pair for the VLC stream */
typedef struct { int cwd[12], len[12], n; int mel_valid, mel_bit; } pairvlc;
static void pair_vlc(const quadsym* q0, const quadsym* q1, int first_row, pairvlc* o)
{
  o->n = 0; o->mel_valid = 0; o->mel_bit = 0;
#define ADD(c, l) do { o->cwd[o->n] = (c); o->len[o->n] = (l); o->n++; } while (0)
  ADD(q0->tuple >> 8, (q0->tuple >> 4) & 7);
  if (q1->exists) ADD(q1->tuple >> 8, (q1->tuple >> 4) & 7);
  int u0 = q0->u, u1 = q1->exists ? q1->u : 0;
  int p0, l0, s0, sl0, x0, xl0, p1, l1, s1, sl1, x1, xl1;
  if (first_row && u0 > 0 && u1 > 0) { o->mel_valid = 1; o->mel_bit = (u0 < u1 ? u0 : u1) > 2; }
  if (first_row && u0 > 2 && u1 > 2) {                              /* :766-772, :1269-1277 */
    uvlc_code(u0 - 2, &p0, &l0, &s0, &sl0, &x0, &xl0); uvlc_code(u1 - 2, &p1, &l1, &s1, &sl1, &x1, &xl1);
    ADD(p0, l0); ADD(p1, l1); ADD(s0, sl0); ADD(s1, sl1); ADD(x0, xl0); ADD(x1, xl1);
  } else if (first_row && u0 > 2 && u1 > 0) {                       /* :773-778, :1278-1284 */
    uvlc_code(u0, &p0, &l0, &s0, &sl0, &x0, &xl0);
    ADD(p0, l0); ADD(u1 - 1, 1); ADD(s0, sl0); ADD(x0, xl0);
  } else {                                                          /* :779-785, :985-988, :1285-1293, :1487-1492 */
    uvlc_code(u0, &p0, &l0, &s0, &sl0, &x0, &xl0); uvlc_code(u1, &p1, &l1, &s1, &sl1, &x1, &xl1);
    ADD(p0, l0); ADD(p1, l1); ADD(s0, sl0); ADD(s1, sl1); ADD(x0, xl0); ADD(x1, xl1);
  }
#undef ADD
}

#define MS_CAP32 ((16384 * 16 + 14) / 15)   /* block_encoder.cpp:550 */
#define MS_CAP64 ((22528 * 16 + 14) / 15)   /* :1041 */
#define MEL_CAP  192
#define VLC_CAP  (3072 - 192)

/* MEL coder over an event list; shared by both variants.  Produces bytes + (tmp, rem) tail. */
static int mel_run_events(melw* mel, const uint8_t* ev, int nev)
{
  int ok = 1;
  for (int i = 0; i < nev; ++i) ok &= mel_event(mel, ev[i]);
  return ok;
}

/* MEL/VLC tail fusion + final assembly (block_encoder.cpp:413-441, 1003-1016) */
static int finish_block(uint8_t* out, int cap, const uint8_t* ms, int ms_len, melw* mel,
                        uint8_t* vlc_last, int* vlc_pos, int vlc_size, int vlc_used, int vlc_tmp)
{
  if (mel->run > 0) if (!mel_bit(mel, 1)) return 0;
  mel->tmp = mel->tmp << mel->rem;
  int mel_mask = (0xFF << mel->rem) & 0xFF;
  int vlc_mask = 0xFF >> (8 - vlc_used);
  if ((mel_mask | vlc_mask) != 0) {
    if (mel->pos >= mel->size) return 0;
    int fuse = mel->tmp | vlc_tmp;
    if ((((fuse ^ mel->tmp) & mel_mask) | ((fuse ^ vlc_tmp) & vlc_mask)) == 0 && fuse != 0xFF &&
        *vlc_pos > 1) {
      mel->buf[mel->pos++] = (uint8_t)fuse;
    } else {
      if (*vlc_pos >= vlc_size) return 0;
      mel->buf[mel->pos++] = (uint8_t)mel->tmp;
      *(vlc_last - *vlc_pos) = (uint8_t)vlc_tmp; (*vlc_pos)++;
    }
  }
  int total = ms_len + mel->pos + *vlc_pos;
  if (total > cap) return 0;
  memcpy(out, ms, (size_t)ms_len);
  memcpy(out + ms_len, mel->buf, (size_t)mel->pos);
  memcpy(out + ms_len + mel->pos, vlc_last - *vlc_pos + 1, (size_t)*vlc_pos);
  int scup = mel->pos + *vlc_pos;
  out[total - 1] = (uint8_t)(scup >> 4);
  out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
  return total;
}

static int ht_encode_core(const uint64_t* buf, int width, int height, int stride, int missing_msbs,
                          uint8_t* out, int cap, int variant, int ms_cap)
{
  build_tables();
  int p = 62 - missing_msbs;
  int QW = (width + 1) >> 1, QH = (height + 1) >> 1, PW = (QW + 1) >> 1;
  const int MS_CAP = ms_cap;

  uint8_t* ms_buf = (uint8_t*)malloc(MS_CAP + 64);
  uint8_t mel_buf[MEL_CAP], vlc_buf[VLC_CAP];
  uint8_t* events = (uint8_t*)malloc((size_t)QW * QH + (size_t)PW * 2 + 16);
  int nev = 0, ok = 1, result = 0;

  melw mel = { mel_buf, 0, MEL_CAP, 8, 0, 0, 0, 1 };
  uint8_t* vlc_last = vlc_buf + VLC_CAP - 1;
  vlcw vlc = { vlc_last, 1, VLC_CAP, 4, 0xF, 1 };      /* vlc_init : block_encoder.cpp:365-375 */
  vlc_last[0] = 0xFF;
  msw ms = { ms_buf, 0, MS_CAP, 8, 0, 0 };
  flatbits fms, fvlc;
  fb_init(&fms, MS_CAP); fb_init(&fvlc, VLC_CAP + 8);
  if (variant) fb_put(&fvlc, 0xF, 4);

  for (int qy = 0; qy < QH && ok; ++qy) {
    int rho_left = 0;
    for (int px = 0; px < PW && ok; ++px) {
      quadsym q0, q1; pairvlc pv;
      quad_symbols(buf, width, height, stride, p, 2 * px, qy, rho_left, &q0);
      quad_symbols(buf, width, height, stride, p, 2 * px + 1, qy, q0.rho, &q1);
      rho_left = q1.rho;
      pair_vlc(&q0, &q1, qy == 0, &pv);
      if (q0.mel_valid) events[nev++] = (uint8_t)q0.mel_bit;
      if (q1.exists && q1.mel_valid) events[nev++] = (uint8_t)q1.mel_bit;
      if (pv.mel_valid) events[nev++] = (uint8_t)pv.mel_bit;
      for (int i = 0; i < pv.n; ++i)
        ok &= variant ? fb_put(&fvlc, (uint32_t)pv.cwd[i], pv.len[i]) : vlc_put(&vlc, pv.cwd[i], pv.len[i]);
      for (int n = 0; n < 4; ++n)
        ok &= variant ? fb_put64(&fms, q0.ms_val[n], q0.ms_len[n]) : ms_put(&ms, q0.ms_val[n], q0.ms_len[n]);
      if (q1.exists)
        for (int n = 0; n < 4; ++n)
          ok &= variant ? fb_put64(&fms, q1.ms_val[n], q1.ms_len[n]) : ms_put(&ms, q1.ms_val[n], q1.ms_len[n]);
    }
  }
  if (!ok) goto done;
  ok &= mel_run_events(&mel, events, nev);
  if (!ok) goto done;

  if (!variant) {
    /* ms_terminate : block_encoder.cpp:517-534 */
    if (ms.used) {
      int t = ms.maxb - ms.used;
      ms.tmp |= (0xFFu & ((1u << t) - 1)) << ms.used;
      if (ms.tmp != 0xFF) { if (ms.pos >= ms.size) goto done; ms.buf[ms.pos++] = (uint8_t)ms.tmp; }
    } else if (ms.maxb == 7) ms.pos--;
    result = finish_block(out, cap, ms.buf, ms.pos, &mel, vlc_last, &vlc.pos, VLC_CAP, vlc.used, vlc.tmp);
  } else {
    /* ---- MagSgn: wave-style stuffing pass over the flat stream (64 "lanes" per step) ---- */
    long T = fms.nbits, pos = 0; int k = 0, prevFF = 0;
    for (;;) {
      int commit = 0, hitFF = 0; long npos = pos;
      for (int lane = 0; lane < 64; ++lane) {
        long start = pos + (lane == 0 ? 0 : (prevFF ? 7 : 8) + 8 * (lane - 1));
        int nb = (lane == 0 && prevFF) ? 7 : 8;
        if (start + nb > T) break;            /* not a full byte: stop */
        uint32_t v = fb_get(&fms, start, nb);
        ms_buf[k + lane] = (uint8_t)v; commit = lane + 1; npos = start + nb;
        if (v == 0xFF) { hitFF = 1; break; } /* lanes beyond were speculated with the wrong phase */
      }
      if (commit == 0) break;
      k += commit; pos = npos; prevFF = hitFF;
      if (k > MS_CAP) goto done;
    }
    int used = (int)(T - pos);
    if (used) {
      int maxb = prevFF ? 7 : 8, t = maxb - used;
      uint32_t tmp = fb_get(&fms, pos, used) | ((0xFFu & ((1u << t) - 1)) << used);
      if (tmp != 0xFF) ms_buf[k++] = (uint8_t)tmp;
    } else if (prevFF) k--;
    /* ---- VLC: backward stuffing pass, 64 lanes per step ---- */
    T = fvlc.nbits; pos = 0; int vpos = 1; int prev = 0xFF;
    for (;;) {
      int commit = 0, special = 0; long npos = pos; int pprev = prev;
      for (int lane = 0; lane < 64; ++lane) {
        long start = pos + 8 * lane;
        if (pprev > 0x8F && start + 7 <= T && fb_get(&fvlc, start, 7) == 0x7F) {
          *(vlc_last - (vpos + lane)) = 0x7F; commit = lane + 1; npos = start + 7; pprev = 0x7F;
          special = 1; break;
        }
        if (start + 8 > T) break;
        uint32_t v = fb_get(&fvlc, start, 8);
        *(vlc_last - (vpos + lane)) = (uint8_t)v; commit = lane + 1; npos = start + 8; pprev = (int)v;
      }
      (void)special;
      if (commit == 0) break;
      vpos += commit; pos = npos; prev = pprev;
      if (vpos >= VLC_CAP) goto done;
    }
    int vused = (int)(T - pos);
    int vtmp = (int)fb_get(&fvlc, pos, vused);
    result = finish_block(out, cap, ms_buf, k, &mel, vlc_last, &vpos, VLC_CAP, vused, vtmp);
  }
done:
  fb_free(&fms); fb_free(&fvlc);
  free(ms_buf); free(events);
  return result;
}

int ojo_ht_encode64(const uint64_t* buf, int width, int height, int stride, int missing_msbs,
                    uint8_t* out, int cap, int variant)
{
  return ht_encode_core(buf, width, height, stride, missing_msbs, out, cap, variant, MS_CAP64);
}

int ojo_ht_encode(const uint32_t* buf, int width, int height, int stride, int missing_msbs,
                  uint8_t* out, int cap, int variant)
{
  uint64_t* wide = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)stride * (size_t)(height > 0 ? height : 1));
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) wide[(size_t)y * stride + x] = (uint64_t)buf[(size_t)y * stride + x] << 32;
  const int r = ht_encode_core(wide, width, height, stride, missing_msbs, out, cap, variant, MS_CAP32);
  free(wide);
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* HT decoder                                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* forward reader with 0xFF -> 7 bit unstuffing, LSB first; `fill` after the end.
 * (block_decoder32.cpp:581-723)                                                               */
static void destuff_forward(flatbits* f, const uint8_t* d, int n, int extra_bytes, int fill)
{
  int unstuff = 0;
  for (int i = 0; i < n + extra_bytes; ++i) {
    int b = i < n ? d[i] : fill;
    int nb = 8 - unstuff;
    fb_put_byte(f, (uint32_t)b, nb);
    unstuff = (b == 0xFF);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* SigProp + MagRef passes (ojph_block_decoder32.cpp:1318-1609), restated on flat bit strings.   */
/* ------------------------------------------------------------------------------------------ */
/* Significance of a 4-row stripe is kept as one 16-bit word per group of 4 columns, column-major:
 * bit 4*c + r = sample (column c, row r) of the group (:1331-1362).  Both passes read the len2
 * bytes that follow the cleanup segment: SigProp forward from its start (LSB first, "after 0xFF
 * only 7 bits", exhausted -> zeros: frwd_read<0> :609-655), MagRef backward from its end (the
 * VLC stuffing rule with unstuff initially true, exhausted -> zeros: rev_read_mrp :453-545). */
static uint32_t next_bit(const flatbits* f, long* pos)      /* exhausted streams feed zeros ... */
{
  /* ... except for the one bit a stuffed last byte may have OR-ed past the end (see fb_put_byte) */
  uint32_t b = *pos <= f->nbits ? fb_get(f, *pos, 1) : 0u;
  ++*pos;
  return b;
}

static void refine_passes(const uint8_t* coded, int lcup, int len2, int num_passes, int p,
                          int width, int height, int stride, uint64_t* out, int stripe_causal,
                          const uint16_t* qinf, int qstr)
{
  int ngroups = (width + 3) >> 2, nstripes = (height + 3) >> 2;
  int mstr = ngroups + 2;
  uint16_t* sigma = (uint16_t*)calloc((size_t)(nstripes + 1) * mstr, 2);
  uint16_t* prev_row = (uint16_t*)calloc((size_t)mstr, 2);
  /* sigma is the quads' rho bits re-arranged (:1331-1351), NOT "the sample is non-zero": a damaged VLC segment may call
   * the samples of a quad's second column / second row significant where the block is one column / row short, and those
   * bits count as neighbours in SigProp and take a bit each in MagRef (their samples are never written here) */
  int QW = (width + 1) >> 1, QH = (height + 1) >> 1;
  for (int qy = 0; qy < QH; ++qy)
    for (int qx = 0; qx < QW; ++qx) {
      const uint32_t rho = ((uint32_t)qinf[(size_t)qy * qstr + qx] >> 4) & 0xFu;
      for (int n = 0; n < 4; ++n)
        if (rho & (1u << n)) {
          const int x = 2 * qx + (n >> 1), y = 2 * qy + (n & 1);
          sigma[(y >> 2) * mstr + (x >> 2)] |= (uint16_t)(1u << (4 * (x & 3) + (y & 3)));
        }
    }

  flatbits fspp, fmrp;
  fb_init(&fspp, len2 + 64); fb_init(&fmrp, len2 + 64);
  destuff_forward(&fspp, coded + lcup, len2, 0, 0);
  { int unstuff = 1;
    for (int i = lcup + len2 - 1; i >= lcup; --i) {
      int b = coded[i];
      int nb = 8 - ((unstuff && (b & 0x7F) == 0x7F) ? 1 : 0);
      fb_put_byte(&fmrp, (uint32_t)b, nb);
      unstuff = b > 0x8F;
    } }
  long spos = 0, mpos = 0;
#define SPP_BIT() next_bit(&fspp, &spos)
#define MRP_BIT() next_bit(&fmrp, &mpos)

  /* ---- significance propagation (:1364-1558) ---- */
  for (int y = 0; y < height; y += 4) {
    uint32_t pattern = 0xFFFFu;
    if (height - y < 4) { pattern = 0x7777u; if (height - y < 3) { pattern = 0x3333u; if (height - y < 2) pattern = 0x1111u; } }
    uint32_t prev = 0;
    uint16_t* cur_sig = sigma + (y >> 2) * mstr;
    uint16_t* nxt_sig = cur_sig + mstr;
    for (int x = 0, g = 0; x < width; x += 4, ++g) {
      int s = x + 4 - width; if (s < 0) s = 0;
      pattern >>= s * 4;
      uint32_t ps = prev_row[g] | ((uint32_t)prev_row[g + 1] << 16);
      uint32_t ns = nxt_sig[g] | ((uint32_t)nxt_sig[g + 1] << 16);
      uint32_t u = (ps & 0x88888888u) >> 3;
      if (!stripe_causal) u |= (ns & 0x11111111u) << 3;
      uint32_t cs = cur_sig[g] | ((uint32_t)cur_sig[g + 1] << 16);
      uint32_t mbr = cs | ((cs & 0x77777777u) << 1) | ((cs & 0xEEEEEEEEu) >> 1) | u;
      uint32_t t = mbr;
      mbr |= (t << 4) | (t >> 4) | (prev >> 12);
      mbr &= pattern; mbr &= ~cs;
      uint32_t new_sig = mbr;
      if (new_sig) {
        static const uint32_t grow[4] = { 0x33u, 0x76u, 0xECu, 0xC8u };
        uint32_t inv_sig = ~cs & pattern;
        for (int c = 0; c < 4; ++c)
          for (int r = 0; r < 4; ++r) {
            uint32_t bit = 1u << (4 * c + r);
            if (!(new_sig & bit)) continue;
            new_sig &= ~bit;
            uint32_t b = SPP_BIT();
            if (b) new_sig |= (grow[r] << (4 * c)) & inv_sig;
          }
        new_sig &= 0xFFFFu;
        for (int c = 0; c < 4; ++c)
          for (int r = 0; r < 4; ++r)
            if (new_sig & (1u << (4 * c + r))) {
              uint32_t sign = SPP_BIT();
              out[(size_t)(y + r) * stride + x + c] = ((uint64_t)sign << 63) | (3ull << (p - 2));
            }
      }
      new_sig |= cs;
      prev_row[g] = (uint16_t)new_sig;
      t = new_sig;
      new_sig |= ((t & 0x7777u) << 1) | ((t & 0xEEEEu) >> 1);
      prev = (new_sig | u) & 0xF000u;
    }
  }
  /* ---- magnitude refinement (:1561-1609): significant samples of the cleanup pass, column-major ---- */
  if (num_passes > 2) {
    uint64_t half = 1ull << (p - 2);
    for (int y = 0; y < height; y += 4)
      for (int x = 0; x < 4 * ngroups; ++x) {
        uint32_t nib = (sigma[(y >> 2) * mstr + (x >> 2)] >> (4 * (x & 3))) & 0xFu;
        for (int r = 0; r < 4; ++r)
          if (nib & (1u << r)) {
            uint32_t sym = MRP_BIT();              /* (a bit is taken for a flagged sample outside the block, too: :1583-1606) */
            if (x < width && y + r < height) out[(size_t)(y + r) * stride + x] ^= ((uint64_t)(1u - sym) << (p - 1)) | half;
          }
      }
  }
#undef SPP_BIT
#undef MRP_BIT
  fb_free(&fspp); fb_free(&fmrp);
  free(sigma); free(prev_row);
}

/* The decoder is written once, with 64-bit samples out (sign in bit 63): ojph_decode_codeblock64
 * (block_decoder64.cpp:766-1660).  ojph_decode_codeblock32 (block_decoder32.cpp:742-1613) is the same algorithm with
 * p32 = 30 - missing_msbs where this uses p64 = 62 - missing_msbs = p32 + 32: its samples are these moved down by 32
 * bits.  `wide` = the 64-bit function's own details:
 *   * its VLC and MagSgn readers take ONE byte at a time and mask the bit a stuffed byte may not carry
 *     (rev_read8 :305-327, frwd_read8 :626-640), where the 32-bit function's four-byte readers OR the whole byte in and
 *     advance by 7 (see fb_put_byte): the same on conforming streams, different on corrupt ones;
 *   * the U-VLC extension: a decoded u above 32 (before the initial row's bias) is followed by 4 more bits, u += 4 ext
 *     (:997-1011, :1119-1133);
 *   * no "32 bits are not enough" checks (:792-826 are commented out there). */
static int ht_decode_core(const uint8_t* coded, int len1, int len2, int num_passes, int missing_msbs,
                          int width, int height, int stride, uint64_t* out, int stripe_causal, int wide)
{
  build_tables();
  if (num_passes > 1 && len2 == 0) num_passes = 1;
  if (num_passes > 3) return 0;
  if (!wide) {
    if (missing_msbs >= 30) return 0;                       /* block_decoder32.cpp:768-789 */
    if (missing_msbs == 29 && num_passes > 1) num_passes = 1;
  } else if (missing_msbs > 61 || (missing_msbs == 61 && num_passes > 1))
    return 0;              /* p = 1 decodes with the cleanup pass alone (block_decoder64.cpp:792-827 has no test); the
                            * refinement passes (3 << (p - 2)) and p = 0 shift by a negative count there: nothing to match */
  int p = 62 - missing_msbs;
  if (len1 < 2) return 0;
  int lcup = len1;
  int scup = ((int)coded[lcup - 1] << 4) + (coded[lcup - 2] & 0xF);
  if (scup < 2 || scup > lcup || scup > 4079) return 0;     /* block_decoder32.cpp:817-819 */

  int QW = (width + 1) >> 1, QH = (height + 1) >> 1;
  int qstr = QW + 3;
  uint16_t* qinf = (uint16_t*)calloc((size_t)(QH + 1) * qstr, 2);
  uint16_t* quq  = (uint16_t*)calloc((size_t)(QH + 1) * qstr, 2);
  uint64_t* vrow = (uint64_t*)calloc((size_t)2 * (2 * QW + 8), 8);
  int ok = 1;

  /* ---- flat, destuffed streams ---- */
  flatbits fmel, fvlc, fms;
  fb_init(&fmel, scup + 64); fb_init(&fvlc, scup + 64); fb_init(&fms, lcup + 64);
  { /* MEL: MSB-first; bytes lcup-scup .. lcup-2, last one |= 0xF, then 0xFF (:93-152,:221-251) */
    int unstuff = 0;
    for (int i = 0; i < scup - 1 + 56; ++i) {
      int b = i < scup - 1 ? coded[lcup - scup + i] : 0xFF;
      if (i == scup - 2) b |= 0xF;
      int nb = 8 - unstuff;
      for (int j = nb - 1; j >= 0; --j) fb_put(&fmel, (uint32_t)(b >> j) & 1, 1);
      unstuff = (b == 0xFF);
    }
  }
  if (!wide) { /* VLC: backward, LSB-first (block_decoder32.cpp:308-405) */
    int d = coded[lcup - 2];
    int t = d >> 4;
    fb_put_byte(&fvlc, (uint32_t)t, 4 - ((t & 7) == 7));   /* rev_init :373-375: all four bits stay in the window, bit 3 of an
                                                              0xF nibble ends up OR-ed with the next byte's LSB */
    int unstuff = (d | 0xF) > 0x8F;
    for (int i = lcup - 3; i >= lcup - scup; --i) {
      int b = coded[i];
      int nb = 8 - ((unstuff && (b & 0x7F) == 0x7F) ? 1 : 0);
      fb_put_byte(&fvlc, (uint32_t)b, nb);
      unstuff = b > 0x8F;
    }
    fb_put(&fvlc, 0, 32); fb_put(&fvlc, 0, 32);
  } else {     /* rev_init8 / rev_read8 (block_decoder64.cpp:305-361): the byte is masked before it is used */
    int val = coded[lcup - 2] >> 4;
    int t = (val & 7) == 7;
    val &= 0xF >> t;
    fb_put(&fvlc, (uint32_t)val, 4 - t);
    int unstuff = val > 0x8;
    for (int i = lcup - 3; i >= lcup - scup; --i) {
      int b = coded[i];
      t = (unstuff && (b & 0x7F) == 0x7F) ? 1 : 0;
      b &= 0xFF >> t;
      fb_put(&fvlc, (uint32_t)b, 8 - t);
      unstuff = b > 0x8F;
    }
    fb_put(&fvlc, 0, 32); fb_put(&fvlc, 0, 32); fb_put(&fvlc, 0, 32);
  }
  if (!wide) destuff_forward(&fms, coded, lcup - scup, 0, 0xFF);
  else {       /* frwd_read8<0xFF> (:626-640) */
    int unstuff = 0;
    for (int i = 0; i < lcup - scup; ++i) {
      int b = coded[i] & (0xFF >> unstuff);
      fb_put(&fms, (uint32_t)b, 8 - unstuff);
      unstuff = (b == 0xFF);
    }
  }

  /* ---- MEL symbol decoder (T.814 decodeMELSym; :170-269 keeps the same state as "runs") ---- */
  long melpos = 0; int mel_k = 0, mel_run = 0, mel_one = 0;
#define VLC_PEEK(n) fb_get(&fvlc, vpos, (n))
  long vpos = 0;

  /* ---- step 1: MEL + VLC + UVLC -> qinf, quq ---- */
  for (int qy = 0; qy < QH; ++qy) {
    uint16_t* row = qinf + (size_t)qy * qstr;
    uint16_t* urow = quq + (size_t)qy * qstr;
    const uint16_t* above = qy ? qinf + (size_t)(qy - 1) * qstr : NULL;
    int tleft = 0;
    for (int qx = 0; qx < QW; qx += 2) {
      int t[2] = { 0, 0 };
      for (int j = 0; j < 2; ++j) {
        int x = qx + j;
        int c_q;
        if (qy == 0)
          c_q = ((tleft & 0x10) << 3) | ((tleft & 0xE0) << 2);              /* :903,:934 */
        else {
          c_q = ((tleft & 0x40) << 2) | ((tleft & 0x80) << 1);              /* :1022,:1059 */
          if (x > 0) c_q |= above[x - 1] & 0x80;                            /* :1024,:1061 */
          c_q |= (above[x] & 0xA0) << 2;                                    /* :990,:1026 */
          c_q |= (above[x + 1] & 0x20) << 4;                                /* :991,:1027 */
        }
        int tv = 0;
        if (x < QW) {
          tv = dec_vlc[qy ? 1 : 0][c_q + (int)VLC_PEEK(7)];
          if (c_q == 0) {
            /* one MEL symbol */
            int sym;
            if (mel_run == 0 && mel_one == 0) {
              int ev = MEL_E[mel_k];
              int bit = (int)fb_get(&fmel, melpos, 1); melpos++;
              if (bit) { mel_run = 1 << ev; mel_k = mel_k < 12 ? mel_k + 1 : 12; }
              else {
                mel_run = 0;
                for (int i = 0; i < ev; ++i) { mel_run = (mel_run << 1) | (int)fb_get(&fmel, melpos, 1); melpos++; }
                mel_k = mel_k > 0 ? mel_k - 1 : 0; mel_one = 1;
              }
            }
            if (mel_run > 0) { mel_run--; sym = 0; } else { mel_one = 0; sym = 1; }
            if (!sym) tv = 0;
          }
          vpos += tv & 7;
        }
        t[j] = tv; row[x] = (uint16_t)tv; tleft = tv;
      }
      /* UVLC */
      int mode = ((t[0] & 0x8) << 3) | ((t[1] & 0x8) << 4);
      int entry, bias0 = 0, bias1 = 0;         /* what the encoder took off u before coding it (uvlc_bias, block_common.cpp:255,290) */
      if (qy == 0) {
        if (mode == 0xC0) {
          int sym;
          if (mel_run == 0 && mel_one == 0) {
            int ev = MEL_E[mel_k];
            int bit = (int)fb_get(&fmel, melpos, 1); melpos++;
            if (bit) { mel_run = 1 << ev; mel_k = mel_k < 12 ? mel_k + 1 : 12; }
            else {
              mel_run = 0;
              for (int i = 0; i < ev; ++i) { mel_run = (mel_run << 1) | (int)fb_get(&fmel, melpos, 1); melpos++; }
              mel_k = mel_k > 0 ? mel_k - 1 : 0; mel_one = 1;
            }
          }
          if (mel_run > 0) { mel_run--; sym = 0; } else { mel_one = 0; sym = 1; }
          if (sym) { mode += 0x40; bias0 = bias1 = 2; }      /* (the one-bit second quad of mode 3 has a bias of 1, but is at most 2) */
        }
        entry = dec_uvlc0[mode + (int)VLC_PEEK(6)];
      } else
        entry = dec_uvlc1[mode + (int)VLC_PEEK(6)];
      vpos += entry & 7; entry >>= 3;
      int len = entry & 0xF;
      int tmp = (int)VLC_PEEK(len);
      vpos += len; entry >>= 4;
      len = entry & 7; entry >>= 3;
      int kap = qy == 0 ? 1 : 0;   /* initial row stores U_q (kappa = 1), others u_q (:971-974,:1082-1085) */
      int u0 = (entry & 7) + (tmp & ~(0xFF << len));
      int u1 = (entry >> 3) + (tmp >> len);
      if (wide) {                  /* the extension of a u above 32 (block_decoder64.cpp:997-1011, :1119-1133) */
        if (u0 - bias0 > 32) { u0 += (int)VLC_PEEK(4) << 2; vpos += 4; }
        if (u1 - bias1 > 32) { u1 += (int)VLC_PEEK(4) << 2; vpos += 4; }
      }
      urow[qx] = (uint16_t)(kap + u0);
      urow[qx + 1] = (uint16_t)(kap + u1);
    }
  }

  /* ---- step 2: MagSgn ---- */
  {
    long mpos = 0;
    int mmsbp2 = missing_msbs + 2;
    uint64_t* vprev = vrow;                 /* v_n of the bottom sample row of the previous quad row */
    uint64_t* vcur = vrow + 2 * QW + 8;
    memset(vprev, 0, sizeof(uint64_t) * (size_t)(2 * QW + 8));
    for (int qy = 0; qy < QH && ok; ++qy) {
      memset(vcur, 0, sizeof(uint64_t) * (size_t)(2 * QW + 8));
      for (int qx = 0; qx < QW; ++qx) {
        int inf = qinf[(size_t)qy * qstr + qx];
        int U_q = quq[(size_t)qy * qstr + qx];
        if (qy > 0) {
          int gamma = inf & 0xF0; gamma &= gamma - 0x10;             /* :1218 */
          /* columns 2qx-1 .. 2qx+2 of the row above (vprev is offset by 1) */
          uint64_t em = vprev[2 * qx] | vprev[2 * qx + 1] | vprev[2 * qx + 2] | vprev[2 * qx + 3];
          int kappa = gamma ? 63 - clz64(em | 2) : 1;                /* :1219-1221 */
          U_q += kappa;
        }
        if (U_q > mmsbp2) { ok = 0; break; }                         /* :1114,:1224 */
        for (int n = 0; n < 4; ++n) {
          int x = 2 * qx + (n >> 1), y = 2 * qy + (n & 1);
          uint64_t val = 0, v_n = 0;
          if (x >= width) break;            /* the second column of an odd-width block's last quad is not read at all, whatever
                                               a damaged VLC segment says about its samples (:1159, :1269; decoder64 :1207, :1316) */
          if (inf & (1 << (4 + n))) {
            int m_n = U_q - ((inf >> (12 + n)) & 1);
            /* beyond the end the stream is all ones */
            uint64_t ms_val = fb_get64(&fms, mpos, 64);
            if (mpos + 64 > fms.nbits) {
              long valid = fms.nbits - mpos; if (valid < 0) valid = 0;
              ms_val |= valid >= 64 ? 0ull : (~0ull << valid);
            }
            mpos += m_n;
            val = ms_val << 63;
            v_n = ms_val & ((m_n < 64 ? (1ull << m_n) : 0ull) - 1ull);
            v_n |= (uint64_t)((inf >> (8 + n)) & 1) << m_n;
            v_n |= 1;
            val |= (v_n + 2) << (p - 1);                             /* :1127-1133 */
          }
          if (x < width && y < height) out[(size_t)y * stride + x] = val;
          if (n & 1) vcur[x + 1] = v_n;
        }
      }
      uint64_t* t = vprev; vprev = vcur; vcur = t;
    }
  }
#undef VLC_PEEK

  if (ok && num_passes > 1)
    refine_passes(coded, lcup, len2, num_passes, p, width, height, stride, out, stripe_causal, qinf, qstr);
  fb_free(&fmel); fb_free(&fvlc); fb_free(&fms);
  free(qinf); free(quq); free(vrow);
  return ok;
}

int ojo_ht_decode64(const uint8_t* coded, int len1, int len2, int num_passes, int missing_msbs,
                    int width, int height, int stride, uint64_t* out, int stripe_causal)
{
  return ht_decode_core(coded, len1, len2, num_passes, missing_msbs, width, height, stride, out, stripe_causal, 1);
}

int ojo_ht_decode(const uint8_t* coded, int len1, int len2, int num_passes, int missing_msbs,
                  int width, int height, int stride, uint32_t* out, int stripe_causal)
{
  uint64_t* wide = (uint64_t*)calloc((size_t)stride * (size_t)(height > 0 ? height : 1), sizeof(uint64_t));
  const int ok = ht_decode_core(coded, len1, len2, num_passes, missing_msbs, width, height, stride, wide, stripe_causal, 0);
  /* (what a failed decode leaves in the buffer is part of the behaviour the tests compare) */
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) out[(size_t)y * stride + x] = (uint32_t)(wide[(size_t)y * stride + x] >> 32);
  free(wide);
  return ok;
}

/* ------------------------------------------------------------------------------------------ */
/* DWT                                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* 1-D 5/3 analysis of n samples with stride st; results written de-interleaved into lo / hi.
 * Restates gen_rev_horz_ana32 (ojph_transform.cpp:336-411); the vertical state machine of
 * resolution::push_line (ojph_resolution.cpp:560-635) computes the same thing per column. */
static void ana53_1d(const int32_t* x, long st, int n, int even, int32_t* lo, long ls, int32_t* hi, long hs)
{
  int lw = (n + (even ? 1 : 0)) >> 1, hw = (n + (even ? 0 : 1)) >> 1;
  if (n == 1) { if (even) lo[0] = x[0]; else hi[0] = x[0] << 1; return; }
  /* predict: hi[i] = x_odd - ((lo_l + lo_r) >> 1) */
  for (int i = 0; i < hw; ++i) {
    int li = i - 1 + (even ? 1 : 0), ri = i + (even ? 1 : 0);
    li = clampi(li, lw);
    ri = clampi(ri, lw);
    int32_t a = x[(long)(2 * li + (even ? 0 : 1)) * st], b = x[(long)(2 * ri + (even ? 0 : 1)) * st];
    hi[i * hs] = x[(long)(2 * i + (even ? 1 : 0)) * st] - ((a + b) >> 1);
  }
  /* update: lo[i] = x_even + ((hi_l + hi_r + 2) >> 2) */
  for (int i = 0; i < lw; ++i) {
    int li = i - 1 + (even ? 0 : 1), ri = i + (even ? 0 : 1);
    li = clampi(li, hw);
    ri = clampi(ri, hw);
    lo[i * ls] = x[(long)(2 * i + (even ? 0 : 1)) * st] + ((2 + hi[li * hs] + hi[ri * hs]) >> 2);
  }
}

/* inverse (gen_rev_horz_syn32, ojph_transform.cpp:514-590) */
static void syn53_1d(int32_t* x, long st, int n, int even, const int32_t* lo, long ls, const int32_t* hi, long hs)
{
  int lw = (n + (even ? 1 : 0)) >> 1, hw = (n + (even ? 0 : 1)) >> 1;
  if (n == 1) { if (even) x[0] = lo[0]; else x[0] = hi[0] >> 1; return; }
  for (int i = 0; i < lw; ++i) {
    int li = i - 1 + (even ? 0 : 1), ri = i + (even ? 0 : 1);
    li = clampi(li, hw);
    ri = clampi(ri, hw);
    x[(long)(2 * i + (even ? 0 : 1)) * st] = lo[i * ls] - ((2 + hi[li * hs] + hi[ri * hs]) >> 2);
  }
  for (int i = 0; i < hw; ++i) {
    int li = i - 1 + (even ? 1 : 0), ri = i + (even ? 1 : 0);
    li = clampi(li, lw);
    ri = clampi(ri, lw);
    int32_t a = x[(long)(2 * li + (even ? 0 : 1)) * st], b = x[(long)(2 * ri + (even ? 0 : 1)) * st];
    x[(long)(2 * i + (even ? 1 : 0)) * st] = hi[i * hs] + ((a + b) >> 1);
  }
}

void ojo_dwt53_fwd(const int32_t* src, int sp, int w, int h, int x_even, int y_even,
                   int32_t* ll, int llp, int32_t* hl, int hlp, int32_t* lh, int lhp,
                   int32_t* hh, int hhp)
{
  int lh_rows = (h + (y_even ? 1 : 0)) >> 1, hh_rows = (h + (y_even ? 0 : 1)) >> 1;
  int32_t* L = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * (lh_rows > 0 ? lh_rows : 1));
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * (hh_rows > 0 ? hh_rows : 1));
  for (int x = 0; x < w; ++x)                       /* vertical first */
    ana53_1d(src + x, sp, h, y_even, L + x, w, H + x, w);
  for (int y = 0; y < lh_rows; ++y)                 /* then horizontal */
    ana53_1d(L + (size_t)y * w, 1, w, x_even, ll + (size_t)y * llp, 1, hl + (size_t)y * hlp, 1);
  for (int y = 0; y < hh_rows; ++y)
    ana53_1d(H + (size_t)y * w, 1, w, x_even, lh + (size_t)y * lhp, 1, hh + (size_t)y * hhp, 1);
  free(L); free(H);
}

void ojo_dwt53_inv(int32_t* dst, int dp, int w, int h, int x_even, int y_even,
                   const int32_t* ll, int llp, const int32_t* hl, int hlp, const int32_t* lh,
                   int lhp, const int32_t* hh, int hhp)
{
  int lh_rows = (h + (y_even ? 1 : 0)) >> 1, hh_rows = (h + (y_even ? 0 : 1)) >> 1;
  int32_t* L = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * (lh_rows > 0 ? lh_rows : 1));
  int32_t* H = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * (hh_rows > 0 ? hh_rows : 1));
  for (int y = 0; y < lh_rows; ++y)                 /* horizontal first */
    syn53_1d(L + (size_t)y * w, 1, w, x_even, ll + (size_t)y * llp, 1, hl + (size_t)y * hlp, 1);
  for (int y = 0; y < hh_rows; ++y)
    syn53_1d(H + (size_t)y * w, 1, w, x_even, lh + (size_t)y * lhp, 1, hh + (size_t)y * hhp, 1);
  for (int x = 0; x < w; ++x)                       /* then vertical */
    syn53_1d(dst + x, dp, h, y_even, L + x, w, H + x, w);
  free(L); free(H);
}

/* 9/7: lifting steps in synthesis order (ojph_params.cpp:2870-2881) */
static const float A97[4] = { (float)0.443506852043971, (float)0.882911075530934,
                              (float)-0.052980118572961, (float)-1.586134342059924 };
static const float K97 = (float)1.230174104914001;

/* 1-D 9/7 analysis without the K scaling; lo/hi hold the de-interleaved signal in place.
 * (gen_irv_horz_ana, ojph_transform.cpp:715-761 ; steps applied from index 3 down to 0,
 * alternately updating the high and the low band.) */
static void lift97_ana(float* lo, long ls, int lw, float* hi, long hs, int hw, int even)
{
  float* lp = lo; long lps = ls; int l_w = lw;
  float* hp = hi; long hps = hs; int h_w = hw;
  int ev = even;
  for (int j = 4; j > 0; --j) {
    float a = A97[j - 1];
    for (int i = 0; i < h_w; ++i) {
      int li = i - 1 + (ev ? 1 : 0), ri = i + (ev ? 1 : 0);
      li = clampi(li, l_w);
      ri = clampi(ri, l_w);
      float s = lp[li * lps] + lp[ri * lps];
      float m = a * s;
      hp[i * hps] = hp[i * hps] + m;
    }
    float* t = lp; lp = hp; hp = t;
    long ts = lps; lps = hps; hps = ts;
    int tw = l_w; l_w = h_w; h_w = tw;
    ev = !ev;
  }
}
static void lift97_syn(float* lo, long ls, int lw, float* hi, long hs, int hw, int even)
{
  /* gen_irv_horz_syn, ojph_transform.cpp:811-830 : aug starts as the low band */
  float* aug = lo; long as = ls; int a_w = lw;
  float* oth = hi; long os = hs; int o_w = hw;
  int ev = even;
  for (int j = 0; j < 4; ++j) {
    float a = A97[j];
    for (int i = 0; i < a_w; ++i) {
      int li = i - 1 + (ev ? 0 : 1), ri = i + (ev ? 0 : 1);
      li = clampi(li, o_w);
      ri = clampi(ri, o_w);
      float s = oth[li * os] + oth[ri * os];
      float m = a * s;
      aug[i * as] = aug[i * as] - m;
    }
    float* t = aug; aug = oth; oth = t;
    long ts = as; as = os; os = ts;
    int tw = a_w; a_w = o_w; o_w = tw;
    ev = !ev;
  }
}

void ojo_dwt97_fwd(const float* src, int sp, int w, int h, int x_even, int y_even,
                   float* ll, int llp, float* hl, int hlp, float* lh, int lhp, float* hh, int hhp)
{
  int lr = (h + (y_even ? 1 : 0)) >> 1, hr = (h + (y_even ? 0 : 1)) >> 1;
  int lw = (w + (x_even ? 1 : 0)) >> 1, hw = (w + (x_even ? 0 : 1)) >> 1;
  const float K = K97, K_inv = 1.0f / K97;
  float* L = (float*)malloc(sizeof(float) * (size_t)w * (lr > 0 ? lr : 1));
  float* H = (float*)malloc(sizeof(float) * (size_t)w * (hr > 0 ? hr : 1));
  /* vertical: de-interleave rows, lift, then H rows *= K, L rows *= 1/K
   * (ojph_resolution.cpp:646-708) */
  for (int y = 0; y < h; ++y) {
    int is_low = ((y & 1) == 0) == (y_even != 0);
    float* d = (is_low ? L : H) + (size_t)(y >> 1) * w;
    memcpy(d, src + (size_t)y * sp, sizeof(float) * (size_t)w);
  }
  if (h > 1) {
    for (int x = 0; x < w; ++x) lift97_ana(L + x, w, lr, H + x, w, hr, y_even);
    for (size_t i = 0; i < (size_t)w * hr; ++i) H[i] *= K;
    for (size_t i = 0; i < (size_t)w * lr; ++i) L[i] *= K_inv;
  } else if (!y_even)
    for (int x = 0; x < w; ++x) H[x] *= 2.0f;
  /* horizontal (gen_irv_horz_ana, ojph_transform.cpp:715-783) */
  for (int pass = 0; pass < 2; ++pass) {
    float* S = pass ? H : L; int rows = pass ? hr : lr;
    float* lo = pass ? lh : ll; int lop = pass ? lhp : llp;
    float* hi = pass ? hh : hl; int hip = pass ? hhp : hlp;
    for (int y = 0; y < rows; ++y) {
      const float* s = S + (size_t)y * w;
      float* l = lo + (size_t)y * lop; float* hgh = hi + (size_t)y * hip;
      if (w == 1) { if (x_even) l[0] = s[0]; else hgh[0] = s[0] * 2.0f; continue; }
      for (int x = 0; x < w; ++x) {
        int is_low = ((x & 1) == 0) == (x_even != 0);
        if (is_low) l[x >> 1] = s[x]; else hgh[x >> 1] = s[x];
      }
      lift97_ana(l, 1, lw, hgh, 1, hw, x_even);
      for (int i = 0; i < lw; ++i) l[i] *= K_inv;
      for (int i = 0; i < hw; ++i) hgh[i] *= K;
    }
  }
  free(L); free(H);
}

void ojo_dwt97_inv(float* dst, int dp, int w, int h, int x_even, int y_even,
                   const float* ll, int llp, const float* hl, int hlp, const float* lh, int lhp,
                   const float* hh, int hhp)
{
  int lr = (h + (y_even ? 1 : 0)) >> 1, hr = (h + (y_even ? 0 : 1)) >> 1;
  int lw = (w + (x_even ? 1 : 0)) >> 1, hw = (w + (x_even ? 0 : 1)) >> 1;
  const float K = K97, K_inv = 1.0f / K97;
  float* L = (float*)malloc(sizeof(float) * (size_t)w * (lr > 0 ? lr : 1));
  float* H = (float*)malloc(sizeof(float) * (size_t)w * (hr > 0 ? hr : 1));
  float* tl = (float*)malloc(sizeof(float) * (size_t)(lw + 1));
  float* th = (float*)malloc(sizeof(float) * (size_t)(hw + 1));
  /* horizontal synthesis of every row (gen_irv_horz_syn, ojph_transform.cpp:786-850) */
  for (int pass = 0; pass < 2; ++pass) {
    float* D = pass ? H : L; int rows = pass ? hr : lr;
    const float* lo = pass ? lh : ll; int lop = pass ? lhp : llp;
    const float* hi = pass ? hh : hl; int hip = pass ? hhp : hlp;
    for (int y = 0; y < rows; ++y) {
      float* d = D + (size_t)y * w;
      if (w == 1) { d[0] = x_even ? lo[(size_t)y * lop] : hi[(size_t)y * hip] * 0.5f; continue; }
      for (int i = 0; i < lw; ++i) tl[i] = lo[(size_t)y * lop + i] * K;
      for (int i = 0; i < hw; ++i) th[i] = hi[(size_t)y * hip + i] * K_inv;
      lift97_syn(tl, 1, lw, th, 1, hw, x_even);
      for (int x = 0; x < w; ++x) {
        int is_low = ((x & 1) == 0) == (x_even != 0);
        d[x] = is_low ? tl[x >> 1] : th[x >> 1];
      }
    }
  }
  /* vertical (ojph_resolution.cpp:833-923): L rows *= K, H rows *= 1/K, then lifting */
  if (h > 1) {
    for (size_t i = 0; i < (size_t)w * lr; ++i) L[i] *= K;
    for (size_t i = 0; i < (size_t)w * hr; ++i) H[i] *= K_inv;
    for (int x = 0; x < w; ++x) lift97_syn(L + x, w, lr, H + x, w, hr, y_even);
  } else if (!y_even)
    for (int x = 0; x < w; ++x) H[x] *= 0.5f;
  for (int y = 0; y < h; ++y) {
    int is_low = ((y & 1) == 0) == (y_even != 0);
    const float* s = (is_low ? L : H) + (size_t)(y >> 1) * w;
    memcpy(dst + (size_t)y * dp, s, sizeof(float) * (size_t)w);
  }
  free(L); free(H); free(tl); free(th);
}

/* ------------------------------------------------------------------------------------------ */
/* DWT, general form: any lifting kernel (ATK), one direction only (DFS), 32- / 64-bit integers or floats */
/* ------------------------------------------------------------------------------------------ */
/* The reference runs every wavelet through the same code: a list of lifting steps in SYNTHESIS order (param_atk,
 * ojph_params.cpp:2870-2896: 5/3 = { (a 1, b 2, e 2), (a -1, b 1, e 1) }, 9/7 = four float steps and K), applied
 *   analysis : steps N-1 .. 0, the first one applied updates the HIGH-pass samples from their low-pass neighbours, the next
 *              one the low-pass samples, and so on (gen_rev_horz_ana32 ojph_transform.cpp:336-412, gen_irv_horz_ana :715-783,
 *              resolution::push_line ojph_resolution.cpp:571-583); reversible x += (b + a (l + r)) >> e, irreversible
 *              x += a (l + r); then, irreversible only, K on one sub-sequence and 1 / K on the other;
 *   synthesis: K first, then steps 0 .. N-1, step 0 updating the LOW-pass samples: x -= ... (gen_rev_horz_syn32 :514-590,
 *              gen_irv_horz_syn :786-852, resolution::pull_line ojph_resolution.cpp:770-783).
 * A missing neighbour at either end is replaced by the one that exists; a sequence of one sample is passed through when it
 * sits at an even coordinate and doubled (analysis) / halved (synthesis) when at an odd one.  (a, b, e) with a = +-1 take
 * special-case branches in the reference (:221-259) that compute the same value as the general formula.  With a DFS marker
 * segment a level may transform one direction only (resolution::pull_line :725-949: HORZ_TRX / VERT_TRX); the other
 * direction's samples are all "low".  Samples are worked on interleaved, in place -- same arithmetic as the reference's
 * split lines.  The horizontal irreversible analysis scales "lp" by 1 / K after its pointers were swapped once per step
 * (:765-777), i.e. the low-pass samples for an even number of steps and the high-pass ones for an odd number: kept. */
#define OJO_GEN_LIFT(T, NAME, IS_FLOAT)                                                                                 \
static void NAME(T* x, long st, int n, int even, const ojo_lift_step* steps, int nsteps, int synthesis)               \
{                                                                                                                      \
  if (n <= 1) return;                                                                                                  \
  for (int k = 0; k < nsteps; ++k) {                                                                                   \
    const int j = synthesis ? k : nsteps - 1 - k;            /* step index */                                          \
    const int tgt_high = synthesis ? (k & 1) : !(k & 1);     /* which sub-sequence this step updates */                \
    const int first = (tgt_high ? (even ? 1 : 0) : (even ? 0 : 1));                                                    \
    for (int t = first; t < n; t += 2) {                                                                               \
      int l = t - 1, r = t + 1;                                                                                        \
      if (l < 0) l = r;                                                                                                \
      if (r >= n) r = l;                                                                                               \
      if (IS_FLOAT) {                                                                                                  \
        const float a = steps[j].A;                                                                                    \
        const float sum = (float)x[l * st] + (float)x[r * st];                                                         \
        const float m = a * sum;                                                                                       \
        x[t * st] = (T)(synthesis ? (float)x[t * st] - m : (float)x[t * st] + m);                                      \
      } else {                                                                                                         \
        /* the reference's own width, wrapping (gen_rev_vert_step32 / 64, ojph_transform.cpp:209-310: si32 / si64 throughout; a */ \
        /* kernel whose gain overflows it is outside the standard, but a damaged ATK segment is read and applied all the same);  */ \
        /* a shift count beyond the width counts modulo it, as the x86 and gfx9 shifters take it */                               \
        if (sizeof(T) == 4) {                                                                                          \
          const uint32_t sum = (uint32_t)x[l * st] + (uint32_t)x[r * st];                                              \
          const int32_t v = (int32_t)((uint32_t)steps[j].b + (uint32_t)steps[j].a * sum) >> (steps[j].e & 31);         \
          x[t * st] = (T)(int32_t)(synthesis ? (uint32_t)x[t * st] - (uint32_t)v : (uint32_t)x[t * st] + (uint32_t)v); \
        } else {                                                                                                       \
          const uint64_t sum = (uint64_t)x[l * st] + (uint64_t)x[r * st];                                              \
          const int64_t v = (int64_t)((uint64_t)(int64_t)steps[j].b + (uint64_t)(int64_t)steps[j].a * sum) >> (steps[j].e & 63); \
          x[t * st] = (T)(int64_t)(synthesis ? (uint64_t)x[t * st] - (uint64_t)v : (uint64_t)x[t * st] + (uint64_t)v); \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }                                                                                                                    \
}
OJO_GEN_LIFT(int32_t, lift_gen_i32, 0)
OJO_GEN_LIFT(int64_t, lift_gen_i64, 0)
OJO_GEN_LIFT(float, lift_gen_f32, 1)

#define OJO_GEN_LEVEL(T, SUF, IS_FLOAT)                                                                                 \
static void dwt_fwd_gen_##SUF(const T* src, int sp, int w, int h, int x_even, int y_even, int horz, int vert,           \
                              const ojo_lift_step* steps, int nsteps, float K,                                          \
                              T* ll, int llp, T* hl, int hlp, T* lh, int lhp, T* hh, int hhp)                           \
{                                                                                                                      \
  T* wk = (T*)malloc(sizeof(T) * (size_t)w * (size_t)h);                                                               \
  for (int y = 0; y < h; ++y) memcpy(wk + (size_t)y * w, src + (size_t)y * sp, sizeof(T) * (size_t)w);                  \
  const float Kinv = 1.0f / K;                                                                                         \
  if (vert) {                                                      /* vertical first (ojph_resolution.cpp:571-601) */  \
    if (h > 1) {                                                                                                       \
      for (int x = 0; x < w; ++x) lift_gen_##SUF(wk + x, w, h, y_even, steps, nsteps, 0);                               \
      if (IS_FLOAT) for (int y = 0; y < h; ++y) {                                                                      \
        const int high = ((y & 1) == 0) != (y_even != 0);                                                              \
        for (int x = 0; x < w; ++x) wk[(size_t)y * w + x] = (T)((float)wk[(size_t)y * w + x] * (high ? K : Kinv));     \
      }                                                                                                                \
    } else if (!y_even) for (int x = 0; x < w; ++x) wk[x] = IS_FLOAT ? (T)((float)wk[x] * 2.0f) : (T)((int64_t)wk[x] * 2); \
  }                                                                                                                    \
  if (horz) for (int y = 0; y < h; ++y) {                                                                              \
    T* row = wk + (size_t)y * w;                                                                                       \
    if (w > 1) {                                                                                                       \
      lift_gen_##SUF(row, 1, w, x_even, steps, nsteps, 0);                                                             \
      if (IS_FLOAT) for (int x = 0; x < w; ++x) {                                                                      \
        const int high = ((x & 1) == 0) != (x_even != 0);                                                              \
        const int lp_is_high = nsteps & 1;                        /* the reference's swapped pointers (:765-777) */    \
        row[x] = (T)((float)row[x] * ((high != lp_is_high) ? K : Kinv));                                               \
      }                                                                                                                \
    } else if (!x_even) row[0] = IS_FLOAT ? (T)((float)row[0] * 2.0f) : (T)((int64_t)row[0] * 2);                       \
  }                                                                                                                    \
  int ly = 0, hy = 0;                                                                                                  \
  for (int y = 0; y < h; ++y) {                                                                                        \
    const int yh = vert ? (((y & 1) == 0) != (y_even != 0)) : 0;                                                       \
    T* lo_row = yh ? lh + (size_t)hy * lhp : ll + (size_t)ly * llp;                                                    \
    T* hi_row = yh ? (hh ? hh + (size_t)hy * hhp : NULL) : (hl ? hl + (size_t)ly * hlp : NULL);                        \
    int lx = 0, hx = 0;                                                                                                \
    for (int x = 0; x < w; ++x) {                                                                                      \
      const int xh = horz ? (((x & 1) == 0) != (x_even != 0)) : 0;                                                     \
      if (xh) hi_row[hx++] = wk[(size_t)y * w + x]; else lo_row[lx++] = wk[(size_t)y * w + x];                         \
    }                                                                                                                  \
    if (yh) hy++; else ly++;                                                                                           \
  }                                                                                                                    \
  free(wk);                                                                                                            \
}                                                                                                                      \
static void dwt_inv_gen_##SUF(T* dst, int dp, int w, int h, int x_even, int y_even, int horz, int vert,                 \
                              const ojo_lift_step* steps, int nsteps, float K,                                          \
                              const T* ll, int llp, const T* hl, int hlp, const T* lh, int lhp, const T* hh, int hhp)   \
{                                                                                                                      \
  T* wk = (T*)malloc(sizeof(T) * (size_t)w * (size_t)h);                                                               \
  const float Kinv = 1.0f / K;                                                                                         \
  int ly = 0, hy = 0;                                                                                                  \
  for (int y = 0; y < h; ++y) {                                                                                        \
    const int yh = vert ? (((y & 1) == 0) != (y_even != 0)) : 0;                                                       \
    const T* lo_row = yh ? lh + (size_t)hy * lhp : ll + (size_t)ly * llp;                                              \
    const T* hi_row = yh ? (hh ? hh + (size_t)hy * hhp : NULL) : (hl ? hl + (size_t)ly * hlp : NULL);                  \
    int lx = 0, hx = 0;                                                                                                \
    for (int x = 0; x < w; ++x) {                                                                                      \
      const int xh = horz ? (((x & 1) == 0) != (x_even != 0)) : 0;                                                     \
      wk[(size_t)y * w + x] = xh ? hi_row[hx++] : lo_row[lx++];                                                        \
    }                                                                                                                  \
    if (yh) hy++; else ly++;                                                                                           \
  }                                                                                                                    \
  if (horz) for (int y = 0; y < h; ++y) {                          /* horizontal first (:740-766) */                   \
    T* row = wk + (size_t)y * w;                                                                                       \
    if (w > 1) {                                                                                                       \
      if (IS_FLOAT) for (int x = 0; x < w; ++x) {                                                                      \
        const int high = ((x & 1) == 0) != (x_even != 0);                                                              \
        row[x] = (T)((float)row[x] * (high ? Kinv : K));                                                               \
      }                                                                                                                \
      lift_gen_##SUF(row, 1, w, x_even, steps, nsteps, 1);                                                             \
    } else if (!x_even) row[0] = IS_FLOAT ? (T)((float)row[0] * 0.5f) : (T)((int64_t)row[0] >> 1);                      \
  }                                                                                                                    \
  if (vert) {                                                                                                          \
    if (h > 1) {                                                                                                       \
      if (IS_FLOAT) for (int y = 0; y < h; ++y) {                                                                      \
        const int high = ((y & 1) == 0) != (y_even != 0);                                                              \
        for (int x = 0; x < w; ++x) wk[(size_t)y * w + x] = (T)((float)wk[(size_t)y * w + x] * (high ? Kinv : K));     \
      }                                                                                                                \
      for (int x = 0; x < w; ++x) lift_gen_##SUF(wk + x, w, h, y_even, steps, nsteps, 1);                               \
    } else if (!y_even) for (int x = 0; x < w; ++x) wk[x] = IS_FLOAT ? (T)((float)wk[x] * 0.5f) : (T)((int64_t)wk[x] >> 1); \
  }                                                                                                                    \
  for (int y = 0; y < h; ++y) memcpy(dst + (size_t)y * dp, wk + (size_t)y * w, sizeof(T) * (size_t)w);                  \
  free(wk);                                                                                                            \
}
OJO_GEN_LEVEL(int32_t, i32, 0)
OJO_GEN_LEVEL(int64_t, i64, 0)
OJO_GEN_LEVEL(float, f32, 1)

void ojo_dwt_fwd_gen(const void* src, int sp, int w, int h, int x_even, int y_even, int elem, int horz, int vert,
                     const ojo_lift_step* steps, int nsteps, float K,
                     void* ll, int llp, void* hl, int hlp, void* lh, int lhp, void* hh, int hhp)
{
  if (elem == 0) dwt_fwd_gen_i32((const int32_t*)src, sp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (int32_t*)ll, llp, (int32_t*)hl, hlp, (int32_t*)lh, lhp, (int32_t*)hh, hhp);
  else if (elem == 1) dwt_fwd_gen_i64((const int64_t*)src, sp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (int64_t*)ll, llp, (int64_t*)hl, hlp, (int64_t*)lh, lhp, (int64_t*)hh, hhp);
  else dwt_fwd_gen_f32((const float*)src, sp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (float*)ll, llp, (float*)hl, hlp, (float*)lh, lhp, (float*)hh, hhp);
}
void ojo_dwt_inv_gen(void* dst, int dp, int w, int h, int x_even, int y_even, int elem, int horz, int vert,
                     const ojo_lift_step* steps, int nsteps, float K,
                     const void* ll, int llp, const void* hl, int hlp, const void* lh, int lhp, const void* hh, int hhp)
{
  if (elem == 0) dwt_inv_gen_i32((int32_t*)dst, dp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (const int32_t*)ll, llp, (const int32_t*)hl, hlp, (const int32_t*)lh, lhp, (const int32_t*)hh, hhp);
  else if (elem == 1) dwt_inv_gen_i64((int64_t*)dst, dp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (const int64_t*)ll, llp, (const int64_t*)hl, hlp, (const int64_t*)lh, lhp, (const int64_t*)hh, hhp);
  else dwt_inv_gen_f32((float*)dst, dp, w, h, x_even, y_even, horz, vert, steps, nsteps, K, (const float*)ll, llp, (const float*)hl, hlp, (const float*)lh, lhp, (const float*)hh, hhp);
}

/* ------------------------------------------------------------------------------------------ */
/* quantise transfer (ojph_codestream_gen.cpp:59-181)                                          */
/* ------------------------------------------------------------------------------------------ */
uint32_t ojo_quant_rev(const int32_t* src, uint32_t* dst, int count, int K_max)
{
  uint32_t shift = 31 - (uint32_t)K_max, mx = 0;
  for (int i = 0; i < count; ++i) {
    int32_t v = src[i];
    uint32_t val = (uint32_t)(v >= 0 ? v : -v) << shift;
    dst[i] = (v >= 0 ? 0u : 0x80000000u) | val; mx |= val;
  }
  return mx;
}
uint32_t ojo_quant_irv(const float* src, uint32_t* dst, int count, float delta_inv)
{
  uint32_t mx = 0;
  for (int i = 0; i < count; ++i) {
    int32_t t = (int32_t)(src[i] * delta_inv);
    uint32_t val = (uint32_t)(t >= 0 ? t : -t);
    dst[i] = (t >= 0 ? 0u : 0x80000000u) | val; mx |= val;
  }
  return mx;
}
void ojo_dequant_rev(const uint32_t* src, int32_t* dst, int count, int K_max)
{
  uint32_t shift = 31 - (uint32_t)K_max;
  for (int i = 0; i < count; ++i) {
    int32_t val = (int32_t)((src[i] & 0x7FFFFFFFu) >> shift);
    dst[i] = (src[i] & 0x80000000u) ? -val : val;
  }
}
void ojo_dequant_irv(const uint32_t* src, float* dst, int count, float delta)
{
  for (int i = 0; i < count; ++i) {
    float val = (float)(src[i] & 0x7FFFFFFFu) * delta;
    dst[i] = (src[i] & 0x80000000u) ? -val : val;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* sample conversion + colour (ojph_colour.cpp:238-571, ojph_tile.cpp:332-518)                 */
/* ------------------------------------------------------------------------------------------ */
void ojo_rev_convert(const int32_t* src, int32_t* dst, int count, int32_t shift)
{ for (int i = 0; i < count; ++i) dst[i] = src[i] + shift; }

void ojo_irv_to_float(const int32_t* src, float* dst, int count, int bit_depth, int is_signed)
{
  float mul = (float)(1.0 / (double)(1ULL << bit_depth));
  int32_t half = is_signed ? 0 : (int32_t)(1ULL << (bit_depth - 1));
  for (int i = 0; i < count; ++i) dst[i] = (float)(src[i] - half) * mul;
}
void ojo_irv_to_int(const float* src, int32_t* dst, int count, int bit_depth, int is_signed)
{
  int32_t neg_limit = (int32_t)INT32_MIN >> (32 - bit_depth);
  float mul = (float)(1ull << bit_depth);
  float up = -(float)neg_limit, low = (float)neg_limit;
  int32_t s_up = INT32_MAX >> (32 - bit_depth), s_low = INT32_MIN >> (32 - bit_depth);
  int32_t half = is_signed ? 0 : (int32_t)(1ULL << (bit_depth - 1));
  for (int i = 0; i < count; ++i) {
    float t = src[i] * mul;
    int32_t v = (int32_t)(t + (t >= 0.0f ? 0.5f : -0.5f));
    v = t >= low ? v : s_low;
    v = t < up ? v : s_up;
    dst[i] = v + half;
  }
}
void ojo_rct_fwd(const int32_t* r, const int32_t* g, const int32_t* b, int32_t* y, int32_t* cb,
                 int32_t* cr, int count)
{
  for (int i = 0; i < count; ++i) {
    int32_t rr = r[i], gg = g[i], bb = b[i];
    y[i] = (rr + (gg << 1) + bb) >> 2; cb[i] = bb - gg; cr[i] = rr - gg;
  }
}
void ojo_rct_inv(const int32_t* y, const int32_t* cb, const int32_t* cr, int32_t* r, int32_t* g,
                 int32_t* b, int count)
{
  for (int i = 0; i < count; ++i) {
    int32_t gg = y[i] - ((cb[i] + cr[i]) >> 2);
    int32_t rr = cr[i] + gg, bb = cb[i] + gg;
    r[i] = rr; g[i] = gg; b[i] = bb;
  }
}
/* ICT constants, ojph_colour.cpp:221-231 */
#define ALPHA_RF 0.299f
#define ALPHA_GF 0.587f
#define ALPHA_BF 0.114f
void ojo_ict_fwd(const float* r, const float* g, const float* b, float* y, float* cb, float* cr,
                 int count)
{
  const float beta_cb = (float)(0.5 / (1 - (double)ALPHA_BF));
  const float beta_cr = (float)(0.5 / (1 - (double)ALPHA_RF));
  for (int i = 0; i < count; ++i) {
    float yy = ALPHA_RF * r[i] + ALPHA_GF * g[i] + ALPHA_BF * b[i];
    float cbb = beta_cb * (b[i] - yy), crr = beta_cr * (r[i] - yy);
    y[i] = yy; cb[i] = cbb; cr[i] = crr;
  }
}
void ojo_ict_inv(const float* y, const float* cb, const float* cr, float* r, float* g, float* b,
                 int count)
{
  const float g_cb2g = (float)(2.0 * (double)ALPHA_BF * (1.0 - (double)ALPHA_BF) / (double)ALPHA_GF);
  const float g_cr2g = (float)(2.0 * (double)ALPHA_RF * (1.0 - (double)ALPHA_RF) / (double)ALPHA_GF);
  const float g_cb2b = (float)(2.0 * (1.0 - (double)ALPHA_BF));
  const float g_cr2r = (float)(2.0 * (1.0 - (double)ALPHA_RF));
  for (int i = 0; i < count; ++i) {
    float gg = y[i] - g_cr2g * cr[i] - g_cb2g * cb[i];
    float rr = y[i] + g_cr2r * cr[i];
    float bb = y[i] + g_cb2b * cb[i];
    r[i] = rr; g[i] = gg; b[i] = bb;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* 64-bit sample path (precision > 32 bits, ojph_params.cpp:1684-1706): transfers, conversion, RCT */
/* ------------------------------------------------------------------------------------------ */
uint64_t ojo_quant_rev64(const int64_t* src, uint64_t* dst, int count, int K_max)   /* gen_rev_tx_to_cb64, ojph_codestream_gen.cpp:81-100 */
{
  uint64_t mx = 0; const int shift = 63 - K_max;
  for (int i = 0; i < count; ++i) {
    const int64_t v = src[i];
    const uint64_t val = (uint64_t)(v >= 0 ? v : -v) << shift;
    dst[i] = (v >= 0 ? 0ull : 0x8000000000000000ull) | val;
    mx |= val;
  }
  return mx;
}
void ojo_dequant_rev64(const uint64_t* src, int64_t* dst, int count, int K_max)     /* gen_rev_tx_from_cb64, :140-153 */
{
  const int shift = 63 - K_max;
  for (int i = 0; i < count; ++i) {
    const int64_t val = (int64_t)((src[i] & 0x7FFFFFFFFFFFFFFFull) >> shift);
    dst[i] = (src[i] >> 63) ? -val : val;
  }
}
/* gen_rev_convert, 32-bit image samples <-> 64-bit lines (ojph_colour.cpp:250-268); nlt3: gen_rev_convert_nlt_type3 (:288-311) */
void ojo_rev_convert_to64(const int32_t* src, int64_t* dst, int count, int64_t shift, int nlt3)
{
  for (int i = 0; i < count; ++i) { const int64_t v = src[i]; dst[i] = nlt3 ? (v >= 0 ? v : -v - shift) : v + shift; }
}
void ojo_rev_convert_from64(const int64_t* src, int32_t* dst, int count, int64_t shift, int nlt3)
{
  for (int i = 0; i < count; ++i) { const int64_t v = src[i]; dst[i] = (int32_t)(nlt3 ? (v >= 0 ? v : -v - shift) : v + shift); }
}
/* gen_rct_forward / backward with 64-bit Y, Cb, Cr and 32-bit R, G, B (ojph_colour.cpp:467-489, :517-541) */
void ojo_rct_fwd64(const int32_t* r, const int32_t* g, const int32_t* b, int64_t* y, int64_t* cb, int64_t* cr, int count)
{
  for (int i = 0; i < count; ++i) {
    const int64_t rr = r[i], gg = g[i], bb = b[i];
    y[i] = (rr + (gg << 1) + bb) >> 2; cb[i] = bb - gg; cr[i] = rr - gg;
  }
}
void ojo_rct_inv64(const int64_t* y, const int64_t* cb, const int64_t* cr, int32_t* r, int32_t* g, int32_t* b, int count)
{
  for (int i = 0; i < count; ++i) {
    const int64_t gg = y[i] - ((cb[i] + cr[i]) >> 2);
    r[i] = (int32_t)(cr[i] + gg); g[i] = (int32_t)gg; b[i] = (int32_t)(cb[i] + gg);
  }
}
