// openjph_amd/csrc/ojph_plan.cpp -- see ojph_plan.h
#include "ojph_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ojphgpu {

// BIBO gains of the 5/3 kernel and sqrt energy gains of the 9/7 kernel, index = number of
// decompositions.  These are properties of the wavelet filters; the values (5 significant
// digits) are the ones the reference quantiser is built on (ojph_params.cpp:513-596), and
// the step sizes written into QCD -- hence the codestream bytes -- depend on them.
static const float bibo53_l[34] = { 1.0000e+00f, 1.5000e+00f, 1.6250e+00f, 1.6875e+00f,
  1.6963e+00f, 1.7067e+00f, 1.7116e+00f, 1.7129e+00f, 1.7141e+00f, 1.7145e+00f, 1.7151e+00f,
  1.7152e+00f, 1.7155e+00f, 1.7155e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
  1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
  1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
  1.7156e+00f, 1.7156e+00f };
static const float bibo53_h[34] = { 2.0000e+00f, 2.5000e+00f, 2.7500e+00f, 2.8047e+00f,
  2.8198e+00f, 2.8410e+00f, 2.8558e+00f, 2.8601e+00f, 2.8628e+00f, 2.8656e+00f, 2.8662e+00f,
  2.8667e+00f, 2.8669e+00f, 2.8670e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
  2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
  2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
  2.8671e+00f, 2.8671e+00f };
static const float energy97_l[34] = { 1.0000e+00f, 1.4021e+00f, 2.0304e+00f, 2.9012e+00f,
  4.1153e+00f, 5.8245e+00f, 8.2388e+00f, 1.1652e+01f, 1.6479e+01f, 2.3304e+01f, 3.2957e+01f,
  4.6609e+01f, 6.5915e+01f, 9.3217e+01f, 1.3183e+02f, 1.8643e+02f, 2.6366e+02f, 3.7287e+02f,
  5.2732e+02f, 7.4574e+02f, 1.0546e+03f, 1.4915e+03f, 2.1093e+03f, 2.9830e+03f, 4.2185e+03f,
  5.9659e+03f, 8.4371e+03f, 1.1932e+04f, 1.6874e+04f, 2.3864e+04f, 3.3748e+04f, 4.7727e+04f,
  6.7496e+04f, 9.5454e+04f };
static const float energy97_h[34] = { 1.4425e+00f, 1.9669e+00f, 2.8839e+00f, 4.1475e+00f,
  5.8946e+00f, 8.3472e+00f, 1.1809e+01f, 1.6701e+01f, 2.3620e+01f, 3.3403e+01f, 4.7240e+01f,
  6.6807e+01f, 9.4479e+01f, 1.3361e+02f, 1.8896e+02f, 2.6723e+02f, 3.7792e+02f, 5.3446e+02f,
  7.5583e+02f, 1.0689e+03f, 1.5117e+03f, 2.1378e+03f, 3.0233e+03f, 4.2756e+03f, 6.0467e+03f,
  8.5513e+03f, 1.2093e+04f, 1.7103e+04f, 2.4187e+04f, 3.4205e+04f, 4.8373e+04f, 6.8410e+04f,
  9.6747e+04f, 1.3682e+05f };

static inline uint32_t div_ceil(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
static inline uint32_t ilog2(uint32_t v) { uint32_t r = 0; while ((1u << (r + 1)) <= v && r < 31) ++r; return r; }

// Visual weights of the qfactor mode, per colour format and component type: levels 1..6 x (HH, LH|HL,
// LH|HL ordering of the reference's table) + LL (ojph_params.cpp:600-793, visual_weights)
enum { VW_420 = 1, VW_422 = 2, VW_444 = 3, VW_ERR = 4 };
static const float vw_cb420[19] = { 0.2724f, 0.5128f, 0.5128f, 0.6692f, 0.9382f, 0.9382f, 1.0888f, 1.3046f, 1.3046f, 1.4156f, 1.5594f,
                                    1.5594f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f };
static const float vw_cr420[19] = { 0.5196f, 0.8260f, 0.8260f, 1.0080f, 1.2928f, 1.2928f, 1.4440f, 1.6508f, 1.6508f, 1.7538f, 1.8848f,
                                    1.8848f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f, 2.0f };
static const float vw_cb422[19] = { 0.1220f, 0.1220f, 0.3626f, 0.3626f, 0.3626f, 0.6634f, 0.6634f, 0.6634f, 0.9225f, 0.9225f, 0.9225f,
                                    1.1027f, 1.1027f, 1.1027f, 1.4142f, 1.4142f, 1.4142f, 1.4142f, 1.4142f };
static const float vw_cr422[19] = { 0.2595f, 0.2595f, 0.5841f, 0.5841f, 0.5841f, 0.9141f, 0.9141f, 0.9141f, 1.1673f, 1.1673f, 1.1673f,
                                    1.3328f, 1.3328f, 1.3328f, 1.4142f, 1.4142f, 1.4142f, 1.4142f, 1.4142f };
static const float vw_cb444[19] = { 0.0263f, 0.0863f, 0.0863f, 0.1362f, 0.2564f, 0.2564f, 0.3346f, 0.4691f, 0.4691f, 0.5444f, 0.6523f,
                                    0.6523f, 0.7078f, 0.7797f, 0.7797f, 1.0f, 1.0f, 1.0f, 1.0f };
static const float vw_cr444[19] = { 0.0773f, 0.1835f, 0.1835f, 0.2598f, 0.4130f, 0.4130f, 0.5040f, 0.6464f, 0.6464f, 0.7220f, 0.8254f,
                                    0.8254f, 0.8769f, 0.9424f, 0.9424f, 1.0f, 1.0f, 1.0f, 1.0f };
static const float vw_y[19] = { 0.0901f, 0.2758f, 0.2758f, 0.7018f, 0.8378f, 0.8378f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f,
                                1.0f, 1.0f, 1.0f, 1.0f, 1.0f };
static const float vw_none[19] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };

static float vw_get(const float* v, uint32_t level, uint32_t band)      // visual_weights::get_weight (:660-672)
{
  if (band == 0) return v[18];
  level = std::min<uint32_t>(level, 6);
  return v[(level - 1) * 3 + (3 - band)];
}

// BIBO gains of a lifting kernel an ATK marker segment describes: the largest factor by which one analysis level can
// raise a low-pass (gl) / a high-pass (gh) sample over the samples it is made from -- the L1 norm of the level's analysis
// filters, found by running the steps (analysis order: N-1 .. 0, the first one on the high-pass samples, K on the high and
// 1 / K on the low ones) over the unit impulses of a 64-sample line.  The reference only READS such kernels and takes the
// magnitude bits of their sub-bands from the QCD / QCC it is given; the writer here has to choose them, and the 5/3 and 9/7
// tables it uses for the Part-1 wavelets say nothing about a kernel with other gains: coefficients past K_max bits would be
// coded with their top bits cut off.
static void atk_bibo_gains(const AtkDef& a, double& gl, double& gh, double* dc_low = nullptr)
{
  const int N = 64;
  std::vector<std::vector<double>> m(N, std::vector<double>(N, 0.0));
  for (int i = 0; i < N; ++i) m[i][i] = 1.0;
  const int ns = (int)a.steps.size();
  for (int k = 0; k < ns; ++k) {
    const ojphgpu_lift_step& st = a.steps[(size_t)(ns - 1 - k)];
    const double c = a.rev ? (double)st.a / std::ldexp(1.0, (int)st.e) : (double)st.A;
    const int first = (k & 1) ? 0 : 1;                       // k = 0 updates the high-pass (odd) samples of a line that starts even
    std::vector<std::vector<double>> nm = m;
    for (int t = first; t < N; t += 2) {
      const int l = t == 0 ? 1 : t - 1, r = t + 1 >= N ? t - 1 : t + 1;
      for (int j = 0; j < N; ++j) nm[t][j] = m[t][j] + c * (m[l][j] + m[r][j]);
    }
    m.swap(nm);
  }
  gl = gh = 0.0;
  if (dc_low) *dc_low = 0.0;
  for (int t = N / 4; t < 3 * N / 4; ++t) {                  // (samples away from the borders: the extension only repeats taps)
    double n1 = 0.0, dc = 0.0;
    for (int j = 0; j < N; ++j) { n1 += std::fabs(m[t][j]); dc += m[t][j]; }
    // (K: the low-pass samples are scaled by 1 / K, the high-pass ones by K -- the other way round in the reference's horizontal
    //  analysis after an odd number of steps, so then the larger of the two)
    const double k_lo = a.rev ? 1.0 : (ns & 1) ? std::max((double)a.K, 1.0 / (double)a.K) : 1.0 / (double)a.K;
    const double k_hi = a.rev ? 1.0 : (ns & 1) ? std::max((double)a.K, 1.0 / (double)a.K) : (double)a.K;
    if (dc_low && !(t & 1)) *dc_low = std::max(*dc_low, std::fabs(dc) * k_lo);   // what a level makes of a constant
    n1 *= (t & 1) ? k_hi : k_lo;
    (t & 1 ? gh : gl) = std::max(t & 1 ? gh : gl, n1);
  }
  if (a.rev) { gl += 0.5 * ns; gh += 0.5 * ns; }            // every step rounds: half a unit each, against small samples
  gl = std::max(gl, 1.0); gh = std::max(gh, 1.0);
}

// One QCD / QCC: param_qcd::make_quant_steps for component `comp` (ojph_params.cpp:1434-1460) with
// set_rev_quant (:1495-1540) or set_irrev_quant (:1542-1599).  ctype 0 Y, 1 Cb, 2 Cr; qfactor 0 = unset.
// `st` = the COD (for the QCD) or the component's own coding style; `base` = the base step handed down
// (<= 0: not set, 2^-min(16, depth) of this component, :1451-1455); the step actually used comes back in it.
static bool make_quant(const Plan& plan, const CodStyle& st, uint32_t comp, uint32_t qfactor, uint32_t ctype, QuantSet& q,
                       std::string& err, float& base)
{
  const ojphgpu_params& p = plan.p;
  const uint32_t D = st.L, depth = plan.comps[comp].bit_depth;
  q.q8.clear(); q.q16.clear();
  // an ATK kernel: cumulative gains from its own filters -- low-pass after d levels gl^d, high-pass of level d gl^(d-1) gh
  // (upper bounds: the product of the levels' norms) -- in place of the 5/3 table
  const AtkDef* atk = plan.atk_of(comp);
  double agl = 1.0, agh = 1.0;
  if (atk) atk_bibo_gains(*atk, agl, agh);
  auto bibo_l = [&](uint32_t d) { return atk ? std::pow(agl, (double)d) : (double)bibo53_l[d]; };
  auto bibo_h = [&](uint32_t d) { return atk ? std::pow(agl, (double)d) * agh : (double)bibo53_h[d]; };
  if (st.rev) {
    uint32_t B = depth + ((comp < 3 && p.color_transform) ? 1 : 0);
    std::vector<uint32_t> e;
    double bl = bibo_l(D);
    uint32_t X = (uint32_t)std::ceil(std::log(bl * bl) / M_LN2);
    e.push_back(B + X);
    uint32_t mx = B + X;
    for (uint32_t d = D; d > 0; --d) {
      double l = bibo_l(d), h = bibo_h(d - 1);
      // (a DFS marker segment: the level has one sub-band, or none -- param_dfs::get_subband_idx; the reference never writes
      // such a QCC, these are the exponents of the two-directional level, an upper bound)
      const uint32_t kind = plan.level_kind(comp, d);
      X = (uint32_t)std::ceil(std::log(h * l) / M_LN2);
      if (kind == 1) { e.push_back(B + X); e.push_back(B + X); mx = std::max(mx, B + X); }
      else if (kind == 2 || kind == 3) { e.push_back(B + X); mx = std::max(mx, B + X); }
      X = (uint32_t)std::ceil(std::log(h * h) / M_LN2);
      if (kind == 1) { e.push_back(B + X); mx = std::max(mx, B + X); }
    }
    if (mx > 38) { err = "bit depth, colour transform and wavelet need more than 38 bits"; return false; }   // :1520-1525
    int guard = std::max(1, (int)mx - 31);
    q.guard_bits = (uint32_t)guard;
    q.sqcd = (uint8_t)(guard << 5);
    for (uint32_t v : e) q.q8.push_back((uint8_t)((v - guard) << 3));
    return true;
  }
  q.guard_bits = 1;
  if (atk) {
    // The normalised samples lie in [-1/2, 1/2).  What decides whether the quantised magnitudes fit is, as for the 9/7 (which
    // the reference codes with ONE guard bit: its low-pass filter passes a constant unchanged, and natural images stay far
    // below the BIBO bound of the high-pass filters), what the kernel makes of SMOOTH content over the levels -- the DC gain
    // of its low-pass filter, which an arbitrary kernel does not normalise to 1 -- times what one high-pass filter can add.
    double dcl = 1.0, l1, h1;
    atk_bibo_gains(*atk, l1, h1, &dcl);
    dcl = std::max(1.0, dcl);
    const double dm1 = (double)(D ? D - 1 : 0);
    const double g1 = std::pow(dcl, dm1) * std::max(dcl, h1);
    int gb = 1 + (int)std::ceil(std::log(std::max(1.0, g1 * g1 * 0.5)) / M_LN2 - 1e-9);
    if (gb > 7) { err = "the wavelet kernel's gain over these decompositions needs more guard bits than a QCD / QCC can signal"; return false; }
    q.guard_bits = (uint32_t)std::max(1, gb);
  }
  q.sqcd = (uint8_t)((q.guard_bits << 5) | 0x2);
  if (!(base > 0.0f)) {                                  // :1451-1455
    uint32_t t = std::min<uint32_t>(16, depth);
    base = 1.0f / (float)(1 << t);
  }
  float g_c = 1.0f, delta_ref = base, power = 1.0f;
  const float* weights = vw_none;
  if (qfactor) {                                         // :1553-1575
    const CompGeo& g = plan.comps[comp];
    const uint32_t fmt = (g.dx == 2 && g.dy == 2) ? VW_420 : (g.dx == 2 && g.dy == 1) ? VW_422 : (g.dx == 1 && g.dy == 1) ? VW_444 : VW_ERR;
    if (fmt == VW_ERR) { err = "Qfactor can only be used on components with 4:4:4, 4:2:2 or 4:2:0 sampling"; return false; }
    if (ctype == 0 && g.dx != 1 && g.dy != 1) { err = "Qfactor can only be used for a Y or luminance component when it is not downsampled."; return false; }
    g_c = ctype == 0 ? 1.0f : ctype == 1 ? 1.8051f / 1.7321f : 1.5734f / 1.7321f;
    // visual_weights::get_delta_ref (:690-722)
    const float t0 = 65, t1 = 97, a0 = 0.04f, a1 = 0.10f;
    const float m_t0 = 2.0f * (1.0f - t0 / 100.0f), m_t1 = 2.0f * (1.0f - t1 / 100.0f);
    const float m_q = qfactor < 50 ? 50.0f / (float)qfactor : 2.0f * (1.0f - (float)qfactor / 100.0f);
    float alpha_q;
    if (qfactor <= 65) { power = 1.0f; alpha_q = a0; }
    else if (qfactor < 97) {
      power = std::log(m_q) - std::log(m_t1);
      power /= std::log(m_t0) - std::log(m_t1);
      alpha_q = a1 * std::pow(a0 / a1, power);
    } else { power = 0.0f; alpha_q = a1; }
    const float eps = std::sqrt(0.5f) * std::ldexp(1.0f, -(int)depth);
    delta_ref = alpha_q * m_q + eps;
    weights = ctype == 0 ? vw_y : ctype == 1 ? (fmt == VW_420 ? vw_cb420 : fmt == VW_422 ? vw_cb422 : vw_cb444)
                                             : (fmt == VW_420 ? vw_cr420 : fmt == VW_422 ? vw_cr422 : vw_cr444);
  }
  auto enc = [&](float delta) {                          // encode_SPqcd :1602-1613
    int exp = 0;
    while (delta < 1.0f) { exp++; delta *= 2.0f; }
    int mant = (int)std::round(delta * (float)(1 << 11)) - (1 << 11);
    mant = mant < (1 << 11) ? mant : 0x7FF;
    q.q16.push_back((uint16_t)((exp << 11) | mant));
  };
  float gl = energy97_l[D];
  float w_b = std::pow(vw_get(weights, D, 0), power);
  enc(delta_ref / (gl * gl * g_c * w_b));
  for (uint32_t d = D; d > 0; --d) {
    float l = energy97_l[d], h = energy97_h[d - 1];
    const uint32_t kind = plan.level_kind(comp, d);        // (DFS: see the reversible branch)
    if (kind == 1 || kind == 2) { w_b = std::pow(vw_get(weights, d, 1), power); enc(delta_ref / (h * l * g_c * w_b)); }
    if (kind == 1 || kind == 3) { w_b = std::pow(vw_get(weights, d, 2), power); enc(delta_ref / (l * h * g_c * w_b)); }
    if (kind == 1) { w_b = std::pow(vw_get(weights, d, 3), power); enc(delta_ref / (h * h * g_c * w_b)); }
  }
  return true;
}

// param_qcd::check_validity (ojph_params.cpp:1359-1432).  The QCD is made with the COD's style for the
// first component that has no COC (component 0 when all have one).  With a qfactor every component
// gets a QCC (Y / Cb / Cr weights for the first three of >= 3 components, Y otherwise); without, a
// component gets one when its decompositions, bit depth, signedness or wavelet differ from what the
// QCD was made for (is_qcc_needed, :1470-1481); such a QCC inherits the QCD's base step (:1426).
bool derive_quant(Plan& plan)
{
  ojphgpu_params& p = plan.p;
  const uint32_t nc = p.num_comps, qf = p.reserved[2];
  plan.qcc.assign(nc, QuantSet());
  plan.qcc_order.clear();
  // QCCs the user made (param_qcd::set_qfactor(comp_idx, ..)), in creation order
  std::vector<uint32_t> user;
  for (uint32_t c = 0; c < OJPHGPU_MAX_COC_COMPS; ++c) {
    if (c >= nc || p.qcc_qfactor[c] == 0) { p.qcc_qfactor[c] = p.qcc_ctype[c] = p.qcc_rank[c] = 0; continue; }
    if (p.qcc_qfactor[c] > 100 || p.qcc_ctype[c] > 2) { plan.error = "Qfactor must be between 1 and 100, the component type Y, Cb or Cr"; return false; }
    user.push_back(c);
  }
  std::stable_sort(user.begin(), user.end(), [&](uint32_t a, uint32_t b) { return p.qcc_rank[a] < p.qcc_rank[b]; });
  for (uint32_t c : user) { plan.qcc[c].present = true; plan.qcc_order.push_back(c); }
  uint32_t qcd_comp = 0;
  for (uint32_t c = 0; c < nc; ++c) if (plan.style(c).rank == 0 && !plan.qcc[c].present) { qcd_comp = c; break; }
  if (qf) for (uint32_t c = 0; c < nc; ++c) if (!plan.qcc[c].present) { plan.qcc[c].present = true; plan.qcc_order.push_back(c); }
  float qcd_base = p.qstep > 0.0f ? p.qstep : -1.0f;
  if (!make_quant(plan, plan.cod, qcd_comp, qf, 0, plan.qcd, plan.error, qcd_base)) return false;   // a reversible COD leaves the base unset
  for (uint32_t c = 0; c < nc; ++c) {
    const CodStyle& st = plan.style(c);
    float base = -1.0f;
    const bool own = c < OJPHGPU_MAX_COC_COMPS && p.qcc_qfactor[c] != 0;
    if (!plan.qcc[c].present) {
      if (st.L == plan.cod.L && st.rev == plan.cod.rev && plan.comps[c].bit_depth == plan.comps[qcd_comp].bit_depth &&
          plan.comps[c].is_signed == plan.comps[qcd_comp].is_signed && st.dfs < 0) continue;    // (DFS: another list of sub-bands)
      plan.qcc[c].present = true; plan.qcc_order.push_back(c);
      base = qcd_base;
    }
    const uint32_t cqf = own ? p.qcc_qfactor[c] : qf;
    const uint32_t ctype = own ? p.qcc_ctype[c] : ((qf && nc >= 3 && c < 3) ? c : 0);
    if (!make_quant(plan, st, c, cqf, ctype, plan.qcc[c], plan.error, base)) return false;
  }
  return true;
}

bool derive_nlt(Plan& plan, bool parsed)
{
  ojphgpu_params& p = plan.p;
  const uint32_t nc = p.num_comps;
  plan.nlt.clear(); plan.nlt3.assign(nc, 0); plan.any_nlt3 = false;
  auto valid = [](uint8_t v) { return v == 0 || v == 1 || v == 4; };
  if (!valid(p.nlt_default)) { plan.error = "unsupported non-linearity type"; return false; }
  struct Obj { int comp; bool enabled; uint8_t type, bd; uint32_t rank; };
  Obj D{ -1, p.nlt_default != 0, (uint8_t)(p.nlt_default ? p.nlt_default - 1 : 0), p.nlt_bd_default, 0 };
  std::vector<Obj> objs;                                   // per-component entries, in creation order
  for (uint32_t c = 0; c < OJPHGPU_MAX_COC_COMPS; ++c) {
    if (!valid(p.nlt_comp[c])) { plan.error = "unsupported non-linearity type"; return false; }
    if (c >= nc || p.nlt_comp[c] == 0) { p.nlt_comp[c] = p.nlt_rank[c] = p.nlt_bd[c] = 0; continue; }   // (:2320-2331 trims what does not exist)
    objs.push_back(Obj{ (int)c, true, (uint8_t)(p.nlt_comp[c] - 1), p.nlt_bd[c], p.nlt_rank[c] });
  }
  std::stable_sort(objs.begin(), objs.end(), [](const Obj& a, const Obj& b) { return a.rank < b.rank; });
  auto obj = [&](uint32_t c) -> Obj* { for (Obj& o : objs) if (o.comp == (int)c) return &o; return nullptr; };
  auto bd_of = [&](uint32_t c) { return (uint8_t)((plan.comps[c].bit_depth - 1) | (plan.comps[c].is_signed ? 0x80 : 0)); };
  const bool any = D.enabled || !objs.empty();
  if (any && !parsed) {                                     // check_validity, on the way to write_headers
    if (D.enabled && D.type == 0) D.enabled = false;
    if (D.enabled && D.type == 3) {
      bool all_same = true; int first = -1;
      for (uint32_t c = 0; c < nc; ++c) {
        Obj* o = obj(c);
        if (!o) { if (first >= 0) all_same = all_same && bd_of(c) == bd_of((uint32_t)first); else first = (int)c; }
        else o->bd = bd_of(c);
      }
      if (all_same && first >= 0) D.bd = bd_of((uint32_t)first);
      else if (!all_same) {
        D.enabled = false;
        for (uint32_t c = 0; c < nc; ++c)
          if (!obj(c)) {
            if (c >= OJPHGPU_MAX_COC_COMPS) { plan.error = "NLT entries are supported for components 0..15"; return false; }
            objs.push_back(Obj{ (int)c, true, 3, bd_of(c), 0 });
          }
      }
    } else
      for (Obj& o : objs) o.bd = bd_of((uint32_t)o.comp);
  }
  if (D.enabled) plan.nlt.push_back(NltSeg{ 65535, D.bd, D.type });
  for (const Obj& o : objs) plan.nlt.push_back(NltSeg{ (uint16_t)o.comp, o.bd, o.type });
  for (uint32_t c = 0; c < nc; ++c) {                       // get_nonlinear_transform + the check of ojph_tile.cpp:292-300
    const Obj* o = obj(c);
    const Obj* e = o ? o : (D.enabled ? &D : nullptr);
    if (!e) continue;
    uint32_t bd = (uint32_t)(e->bd & 0x7F) + 1; bd = bd <= 38 ? bd : 38;
    if (bd != plan.comps[c].bit_depth || ((e->bd & 0x80) != 0) != plan.comps[c].is_signed) {
      plan.error = "Mismatch between Ssiz from the SIZ marker segment and BDnlt from an NLT marker segment";
      return false;
    }
    plan.nlt3[c] = e->type == 3 && plan.comps[c].is_signed;
    plan.any_nlt3 |= plan.nlt3[c] != 0;
  }
  return true;
}

// index of a sub-band's entry in its QCD / QCC; with a DFS marker segment the levels below contribute 3, 1 or 0
// sub-bands each (param_dfs::get_subband_idx, ojph_params.cpp:2550-2572)
static inline uint32_t band_index(const Plan& plan, uint32_t comp, uint32_t res, uint32_t band)
{
  if (res == 0) return 0;
  if (plan.style(comp).dfs < 0) return (res - 1) * 3 + band;
  static const uint32_t ns[4] = { 0, 3, 1, 1 };
  const uint32_t L = plan.style(comp).L;
  uint32_t idx = 0;
  for (uint32_t i = 1; i < res; ++i) idx += ns[plan.level_kind(comp, L - i + 1)];
  idx += band;
  if (plan.level_kind(comp, L - res + 1) == 3 && band == 2) --idx;
  return idx;
}

uint32_t band_Kmax(const Plan& plan, uint32_t comp, uint32_t res, uint32_t band)   // ojph_params.cpp:1715-1749
{
  const QuantSet& q = plan.quant(comp);
  uint32_t idx = band_index(plan, comp, res, band);
  if ((q.sqcd & 0x1F) == 0) {                            // the style of the marker segment decides, as in the reference
    idx = std::min<uint32_t>(idx, (uint32_t)q.q8.size() - 1);
    uint32_t nb = q.q8[idx] >> 3;
    nb = nb == 0 ? 0 : nb - 1;
    return nb + q.guard_bits;
  }
  idx = std::min<uint32_t>(idx, (uint32_t)q.q16.size() - 1);
  return (uint32_t)(q.q16[idx] >> 11) - 1 + q.guard_bits;
}

float band_delta(const Plan& plan, uint32_t comp, uint32_t res, uint32_t band)    // ojph_params.cpp:1650-1681
{
  static const float arr[4] = { 1.0f, 2.0f, 2.0f, 4.0f };
  const QuantSet& q = plan.quant(comp);
  uint32_t idx = std::min<uint32_t>(band_index(plan, comp, res, band), (uint32_t)q.q16.size() - 1);
  int eps = q.q16[idx] >> 11;
  float mantissa = (float)((q.q16[idx] & 0x7FF) | 0x800) * arr[band];
  mantissa /= (float)(1 << 11);
  mantissa /= (float)(1u << eps);
  return mantissa;
}

uint32_t block_scratch_bytes(uint32_t w, uint32_t h, uint32_t K_max)
{
  // MagSgn: every sample <= K_max+1 bits, stuffing adds at most 1 bit per 7;
  // MEL + VLC: the reference's fixed 3072-byte budget (ojph_block_encoder.cpp:552-557)
  uint64_t bits = (uint64_t)w * h * (K_max + 2);
  uint64_t ms = (bits + 6) / 7 + 16;
  uint64_t total = ms + 3072 + 64;
  return (uint32_t)((total + 63) & ~63ull);
}

// param_qcd::propose_precision (ojph_params.cpp:1684-1706): the largest K_max of the component's QCD / QCC -- of the three
// colour components' when the colour transform is employed and this is one of them -- plus a sign bit plus one bit the
// block coder wants; above 32 the reference takes its 64-bit sample path (ojph_resolution.cpp:209-229, ojph_subband.cpp:
// 94-110, ojph_codeblock.cpp:57-99).  Only reversible components can take it here: the reference's encoder has no
// irreversible 64-bit transfer (codeblock_fun::init sets tx_to_cb64 = NULL, ojph_codeblock_fun.cpp:174), and its decoder
// scales such samples by a step size computed with a 32-bit shift (ojph_subband.cpp:159) -- nothing to be identical to.
static uint32_t largest_Kmax(const QuantSet& q)                    // param_qcd::get_largest_Kmax (:1751-1775)
{
  uint32_t nb = 0;
  if ((q.sqcd & 0x1F) == 0) for (uint8_t v : q.q8) { const uint32_t t = v >> 3; nb = std::max(nb, t == 0 ? 0u : t - 1u); }
  else for (uint16_t v : q.q16) nb = std::max<uint32_t>(nb, (uint32_t)(v >> 11) - 1u);
  return nb + q.guard_bits;
}

bool derive_precision(Plan& plan)
{
  const uint32_t nc = plan.p.num_comps;
  plan.wide.assign(nc, 0); plan.any_wide = false;
  for (uint32_t c = 0; c < nc; ++c) {
    uint32_t precision = 0;
    if (plan.p.color_transform && c < 3) for (uint32_t i = 0; i < 3; ++i) precision = std::max(precision, largest_Kmax(plan.quant(i)));
    else precision = largest_Kmax(plan.quant(c));
    precision += 2;
    if (!plan.style(c).rev && plan.comps[c].bit_depth > 32) { plan.error = "irreversible coding of samples deeper than 32 bits: not supported"; return false; }   // (ojph_colour.cpp:331 asserts the same)
    if (precision <= 32) continue;
    if (!plan.style(c).rev) { plan.error = "an irreversibly transformed component needs more than 32 bits of precision: not supported"; return false; }
    plan.wide[c] = 1; plan.any_wide = true;
  }
  return true;
}

ojphgpu_lift Plan::lift_of(uint32_t comp, uint32_t d) const
{
  ojphgpu_lift k; memset(&k, 0, sizeof(k));
  const CodStyle& st = style(comp);
  const uint32_t kind = level_kind(comp, d);
  k.horz = kind == 1 || kind == 2; k.vert = kind == 1 || kind == 3;
  k.elem = st.rev ? ((comp < wide.size() && wide[comp]) ? 1u : 0u) : 2u;
  k.K = 1.0f;
  if (const AtkDef* a = atk_of(comp)) {
    k.num_steps = (uint32_t)a->steps.size(); k.K = a->K;
    for (size_t i = 0; i < a->steps.size(); ++i) k.steps[i] = a->steps[i];
  } else if (st.rev) {                                       // param_atk::init_rev53 (ojph_params.cpp:2883-2896)
    k.num_steps = 2;
    k.steps[0].a = 1; k.steps[0].b = 2; k.steps[0].e = 2;
    k.steps[1].a = -1; k.steps[1].b = 1; k.steps[1].e = 1;
  } else {                                                   // param_atk::init_irv97 (:2870-2881)
    k.num_steps = 4; k.K = (float)1.230174104914001;
    k.steps[0].A = (float)0.443506852043971; k.steps[1].A = (float)0.882911075530934;
    k.steps[2].A = (float)-0.052980118572961; k.steps[3].A = (float)-1.586134342059924;
  }
  return k;
}

// Places of the planes in the arena (32-bit elements): per tile-component, resolution by resolution, the raw plane of the
// resolution (r > 0) and then its bands.  A component on the 64-bit sample path has 64-bit samples in all of them: two
// elements per sample (pitch stays in samples; such planes start on even elements).  Then the DWT levels, which quote
// those places.  Called by build_plan, and again by the parser once the codestream's own QCD / QCC are in place.
void assign_planes(Plan& plan)
{
  uint64_t arena = 0;
  auto alloc = [&](uint32_t w, uint32_t h, uint32_t& pitch, bool wide) {
    pitch = (std::max<uint32_t>(w, 1) + 63u) & ~63u;
    uint64_t off = arena;
    arena += ((uint64_t)pitch * std::max<uint32_t>(h, 1) + 64) * (wide ? 2u : 1u);   // +64: kernels may over-read a row tail
    arena = (arena + 63) & ~63ull;
    return off;
  };
  plan.levels.clear();
  for (const Tile& t : plan.tiles)
    for (uint32_t ci : t.comps) {
      const TileComp& tc = plan.tcomps[ci];
      const uint32_t c = tc.comp, L = (uint32_t)tc.res.size() - 1;
      const bool wide = c < plan.wide.size() && plan.wide[c] != 0;
      for (uint32_t r = 0; r <= L; ++r) {
        Resolution& R = plan.ress[tc.res[r]];
        R.plane_off = 0; R.pitch = 0;
        if (r > 0) R.plane_off = alloc(R.r.w, R.r.h, R.pitch, wide);
        for (int b = 0; b < 4; ++b)
          if (R.band[b] >= 0) { Band& B = plan.bands[(size_t)R.band[b]]; B.plane_off = alloc(B.r.w, B.r.h, B.pitch, wide); }
      }
      // DWT levels: res r -> res r-1 (or the LL band when r-1 == 0) + bands of res r
      for (uint32_t r = L; r > 0; --r) {
        const Resolution& R = plan.ress[tc.res[r]];
        const Resolution& C = plan.ress[tc.res[r - 1]];
        ojphgpu_level_info lv; memset(&lv, 0, sizeof(lv));
        lv.tile = t.idx; lv.comp = c; lv.res = r;
        lv.w = R.r.w; lv.h = R.r.h; lv.x_even = (R.r.x0 & 1) == 0; lv.y_even = (R.r.y0 & 1) == 0;
        lv.src_off = R.plane_off; lv.src_pitch = R.pitch; lv.kind = R.kind;
        if (r - 1 == 0) { const Band& B = plan.bands[(size_t)C.band[0]]; lv.ll_off = B.plane_off; lv.ll_pitch = B.pitch; }
        else { lv.ll_off = C.plane_off; lv.ll_pitch = C.pitch; }
        if (R.band[1] >= 0) { const Band& HL = plan.bands[(size_t)R.band[1]]; lv.hl_off = HL.plane_off; lv.hl_pitch = HL.pitch; }
        if (R.band[2] >= 0) { const Band& LH = plan.bands[(size_t)R.band[2]]; lv.lh_off = LH.plane_off; lv.lh_pitch = LH.pitch; }
        if (R.band[3] >= 0) { const Band& HH = plan.bands[(size_t)R.band[3]]; lv.hh_off = HH.plane_off; lv.hh_pitch = HH.pitch; }
        plan.levels.push_back(lv);
      }
    }
  plan.arena_elems = arena;
}

int build_plan(const ojphgpu_params& pin, Plan& plan)
{
  { const bool parsed = plan.parsed, no_packets = plan.no_packets; plan = Plan(); plan.parsed = parsed; plan.no_packets = no_packets; }
  ojphgpu_params p = pin;
  auto fail = [&](const char* m) { plan.error = m; return OJPHGPU_E_INVALID; };
  if (p.width == 0 || p.height == 0 || p.num_comps == 0) return fail("empty image");
  if (p.num_comps > 16384) return fail("too many components");
  if (p.bit_depth < 1 || p.bit_depth > 38) return fail("bit depth must be 1..38");            // Ssiz: 7 bits, T.800 allows up to 38
  if (p.num_decomps > 32) return fail("too many decompositions");
  if (p.block_w == 0) p.block_w = 64;
  if (p.block_h == 0) p.block_h = 64;
  const uint32_t lbw = ilog2(p.block_w), lbh = ilog2(p.block_h);
  auto block_ok = [](uint32_t a, uint32_t b) { return a >= 2 && b >= 2 && a <= 10 && b <= 10 && a + b <= 12; };
  if ((1u << lbw) != p.block_w || (1u << lbh) != p.block_h || !block_ok(lbw, lbh))
    return fail("code-block dimensions must be powers of two, 4..1024, area <= 4096");
  if (p.color_transform && p.num_comps < 3)
    return fail("color transform can only be employed when the image has 3 or more color components");   // ojph_params_local.h:450-453
  if (p.prog_order > 4) return fail("unknown progression order");
  // reference grid: image offset, tile offset, sub-sampling (ojph_params.cpp:88-150 check_validity)
  if ((uint64_t)p.image_x0 + p.width > 0xFFFFFFFFull || (uint64_t)p.image_y0 + p.height > 0xFFFFFFFFull)
    return fail("image extent exceeds 2^32 - 1");
  if (p.tile_x0 > p.image_x0 || p.tile_y0 > p.image_y0) return fail("tile offset has to be smaller than the image offset");
  const uint32_t X1 = p.image_x0 + p.width, Y1 = p.image_y0 + p.height;            // image extent
  if (p.tile_w == 0 || p.tile_h == 0) {                 // not set: one tile, sized as write_headers sizes it
    if ((uint64_t)X1 + p.image_x0 > 0xFFFFFFFFull || (uint64_t)Y1 + p.image_y0 > 0xFFFFFFFFull) return fail("image extent too large");
    p.tile_w = X1 + p.image_x0; p.tile_h = Y1 + p.image_y0;                        // ojph_codestream_local.cpp:562-570
  }
  if ((uint64_t)p.tile_x0 + p.tile_w <= p.image_x0 || (uint64_t)p.tile_y0 + p.tile_h <= p.image_y0)
    return fail("the top left tile must intersect with the image");
  plan.comps.resize(p.num_comps);
  plan.frame_elems = 0;
  bool subsampled = false;
  for (uint32_t c = 0; c < p.num_comps; ++c) {
    CompGeo& g = plan.comps[c];
    g.dx = c < OJPHGPU_MAX_SUBSAMPLED_COMPS && p.comp_dx[c] ? p.comp_dx[c] : 1;
    g.dy = c < OJPHGPU_MAX_SUBSAMPLED_COMPS && p.comp_dy[c] ? p.comp_dy[c] : 1;
    subsampled |= g.dx != 1 || g.dy != 1;
    g.x0 = div_ceil(p.image_x0, g.dx); g.y0 = div_ceil(p.image_y0, g.dy);
    g.w = div_ceil(X1, g.dx) - g.x0; g.h = div_ceil(Y1, g.dy) - g.y0;             // param_siz::get_recon_width (:330-346)
    g.frame_off = plan.frame_elems;
    plan.frame_elems += (uint64_t)g.w * g.h;
    g.bit_depth = c < OJPHGPU_MAX_SUBSAMPLED_COMPS && p.comp_depth[c] ? p.comp_depth[c] : p.bit_depth;
    g.is_signed = c < OJPHGPU_MAX_SUBSAMPLED_COMPS && p.comp_sign[c] ? p.comp_sign[c] == 2 : p.is_signed != 0;
    if (g.bit_depth < 1 || g.bit_depth > 38) return fail("bit depth must be 1..38");
  }
  // nothing beyond what a device could hold is planned: a frame of 2^36 samples is 256 GB of int32
  // coefficients, and block / band indices are 32 bits wide
  if (plan.frame_elems > (1ull << 36)) return fail("frame too large for the device path (more than 2^36 samples)");
  for (uint32_t c = 0; c < OJPHGPU_MAX_SUBSAMPLED_COMPS; ++c) {                    // canonical form: 0 where the default applies
    p.comp_depth[c] = c < p.num_comps && plan.comps[c].bit_depth != p.bit_depth ? (uint8_t)plan.comps[c].bit_depth : 0;
    p.comp_sign[c] = c < p.num_comps && plan.comps[c].is_signed != (p.is_signed != 0) ? (plan.comps[c].is_signed ? 2 : 1) : 0;
  }
  if (p.reserved[2] > 100) return fail("Qfactor must be between 1 and 100");       // ojph_params.cpp:1487
  for (uint32_t c = 0; c < OJPHGPU_MAX_SUBSAMPLED_COMPS; ++c) {                    // canonical form: 1 is stored as 1
    p.comp_dx[c] = c < p.num_comps ? (uint8_t)plan.comps[c].dx : 0;
    p.comp_dy[c] = c < p.num_comps ? (uint8_t)plan.comps[c].dy : 0;
  }
  if (!plan.parsed && (p.prog_order == 2 || p.prog_order == 3))                    // ojph_params_local.h:488-499 (the writer's check)
    for (const CompGeo& g : plan.comps)
      if ((g.dx & (g.dx - 1)) || (g.dy & (g.dy - 1)))
        return fail("For RPCL and PCRL progression orders, component downsampling factors have to be powers of 2");
  if (p.color_transform)                                                           // ojph_codestream_local.cpp:586-597
    for (uint32_t c = 1; c < 3; ++c) {
      if (plan.comps[c].dx != plan.comps[0].dx || plan.comps[c].dy != plan.comps[0].dy)
        return fail("the colour transform needs the first three components to have the same sub-sampling");
      // bit depth and signedness: the WRITER's test (param_cod::check_validity, ojph_params_local.h:455-490).  The reader
      // has none: it converts component by component (ojph_tile.cpp:439-518), and so do the kernels here.
      if (!plan.parsed && (plan.comps[c].bit_depth != plan.comps[0].bit_depth || plan.comps[c].is_signed != plan.comps[0].is_signed))
        return fail("the colour transform needs the first three components to have the same bit depth and signedness");
    }
  (void)subsampled;
  // Part 2: the ATK / DFS marker segments (what the reference's reader accepts, ojph_params.cpp:2596-2644, :2770-2866)
  plan.atks.clear(); plan.dfss.clear();
  for (uint32_t i = 0; i < OJPHGPU_MAX_ATK; ++i) {
    ojphgpu_atk& a = p.atk[i];
    if (a.index == 0) { memset(&a, 0, sizeof(a)); continue; }
    if (a.index < 2) return fail("an ATK marker segment's index must be 2..255");
    for (const AtkDef& o : plan.atks) if (o.index == a.index) return fail("two ATK marker segments with one index");
    if (a.num_steps == 0 || a.num_steps > OJPHGPU_MAX_LIFT_STEPS) return fail("an ATK marker segment with no or too many lifting steps");
    a.reversible = a.reversible ? 1 : 0;
    if (a.reversible ? a.coeff_type > 1 : (a.coeff_type < 2 || a.coeff_type > 3)) return fail("ATK coefficient type does not fit the kernel");
    AtkDef d; d.index = a.index; d.rev = a.reversible != 0; d.coeff_type = a.coeff_type; d.K = a.reversible ? 1.0f : a.K;
    for (uint32_t k = 0; k < a.num_steps; ++k) {
      ojphgpu_lift_step st = a.steps[k];
      if (d.rev) { st.A = 0.0f; if (st.e < 0 || st.e > (plan.parsed ? 255 : 62)) return fail("ATK lifting step with an impossible shift"); }   // (read: whatever the Eatk byte says -- the shifters count modulo the width)
      else { st.a = st.b = st.e = 0; }
      a.steps[k] = st; d.steps.push_back(st);
    }
    for (uint32_t k = a.num_steps; k < OJPHGPU_MAX_LIFT_STEPS; ++k) memset(&a.steps[k], 0, sizeof(a.steps[k]));
    if (!plan.parsed && !d.rev && !(d.K > 0.0f)) return fail("ATK scaling factor must be positive");
    plan.atks.push_back(d);
  }
  for (uint32_t i = 0; i < OJPHGPU_MAX_DFS; ++i) {
    ojphgpu_dfs& f = p.dfs[i];
    if (!f.used) { memset(&f, 0, sizeof(f)); continue; }
    f.used = 1; f.reserved = 0;
    if (f.index > 15) return fail("a DFS marker segment's index must be 0..15");            // :2620-2622
    if (f.num_levels == 0 || f.num_levels > 32) return fail("a DFS marker segment describes 1..32 levels");
    for (const DfsDef& o : plan.dfss) if (o.index == f.index) return fail("two DFS marker segments with one index");
    DfsDef d; d.index = f.index;
    for (uint32_t k = 0; k < f.num_levels; ++k) { if (f.types[k] > 3) return fail("unknown DFS level type"); d.types.push_back(f.types[k]); }
    for (uint32_t k = f.num_levels; k < 32; ++k) f.types[k] = 0;
    plan.dfss.push_back(d);
  }
  auto find_atk = [&](uint32_t idx) -> const AtkDef* { for (const AtkDef& a : plan.atks) if (a.index == idx) return &a; return nullptr; };
  // the COD's style
  plan.cod = CodStyle();
  plan.cod.L = p.num_decomps; plan.cod.lbw = lbw; plan.cod.lbh = lbh; plan.cod.rev = p.reversible != 0;
  plan.cod.causal = (p.reserved[0] & 1u) != 0;
  if (p.wavelet == 1) return fail("the wavelet byte names an ATK marker segment: 2..255");
  plan.cod.wavelet = p.wavelet;
  if (p.wavelet >= 2) {                                      // param_cod::update_atk (ojph_params.cpp:1279-1297)
    const AtkDef* a = find_atk(p.wavelet);
    if (!a) return fail("the COD names an ATK marker segment that is not there");
    plan.cod.rev = a->rev; p.reversible = a->rev ? 1 : 0;
  }
  {
    uint32_t lpw = 15, lph = 15;
    bool per_res = false;                                         // a list of precinct sizes, one per resolution
    for (uint32_t i = 0; i <= p.num_decomps && i < 36; ++i) per_res |= p.precinct_exps[i] != 0;
    if (p.precinct_w && p.precinct_h) {
      lpw = ilog2(p.precinct_w); lph = ilog2(p.precinct_h);
      if ((1u << lpw) != p.precinct_w || (1u << lph) != p.precinct_h || lpw > 15 || lph > 15)
        return fail("precinct size must be a power of two <= 32768");
      if (!per_res && p.num_decomps > 0 && (lpw == 0 || lph == 0)) return fail("precinct size too small");   // (one size for every resolution; a list may start with a 1: ojph_params.cpp:1190-1198)
    }
    if (per_res) {
      if (p.num_decomps >= 36) return fail("too many resolutions for a precinct list");
      for (uint32_t i = 1; i <= p.num_decomps; ++i)
        if ((p.precinct_exps[i] & 15) == 0 || (p.precinct_exps[i] >> 4) == 0) return fail("precinct size too small");   // ojph_params.cpp:208
      p.precinct_w = 1u << (p.precinct_exps[0] & 15); p.precinct_h = 1u << (p.precinct_exps[0] >> 4);
    } else memset(p.precinct_exps, 0, sizeof(p.precinct_exps));
    plan.cod.has_prec = p.precinct_w && p.precinct_h;
    if (plan.cod.has_prec)
      for (uint32_t i = 0; i <= p.num_decomps && i < 36; ++i) plan.cod.pexp[i] = per_res ? p.precinct_exps[i] : (uint8_t)(lpw | (lph << 4));
  }
  // the COCs (components beyond the 16th follow the COD)
  plan.coc.assign(p.num_comps, CodStyle());
  plan.max_decomps = 0;
  for (uint32_t c = 0; c < OJPHGPU_MAX_COC_COMPS; ++c) {
    ojphgpu_coc& k = p.coc[c];
    if (c >= p.num_comps || k.rank == 0) { memset(&k, 0, sizeof(k)); continue; }   // canonical form
    CodStyle& st = plan.coc[c];
    st.rank = k.rank; st.L = k.num_decomps; st.lbw = k.log_block_w; st.lbh = k.log_block_h; st.rev = k.reversible != 0;
    k.reserved[0] &= 0x9Fu;                                                     // bit 0: vertically causal (parser); bit 7 + bits 1..4: DFS
    st.causal = (k.reserved[0] & 1u) != 0;
    if (k.reserved[0] & 0x80u) {                                                // the decomposition comes from a DFS marker segment,
      st.dfs = (k.reserved[0] >> 1) & 0xF;                                      // the number of levels from the COD (ojph_params_local.h:503-516)
      bool found = false;
      for (const DfsDef& f : plan.dfss) found = found || (int)f.index == st.dfs;
      if (!found) return fail("a COC names a DFS marker segment that is not there");   // ojph_resolution.cpp:276-287
      st.L = p.num_decomps; k.num_decomps = (uint8_t)p.num_decomps;
    } else k.reserved[0] &= 1u;
    if (k.reserved[1] == 1) return fail("the wavelet byte names an ATK marker segment: 2..255");
    st.wavelet = k.reserved[1];
    if (st.wavelet >= 2) {
      const AtkDef* a = find_atk(st.wavelet);
      if (!a) return fail("a COC names an ATK marker segment that is not there");
      st.rev = a->rev;
    }
    k.reversible = st.rev ? 1 : 0;
    if (st.L > 32) return fail("too many decompositions");
    if (!block_ok(st.lbw, st.lbh)) return fail("code-block dimensions must be powers of two, 4..1024, area <= 4096");
    st.has_prec = k.has_precincts != 0;
    k.has_precincts = st.has_prec ? 1 : 0;
    if (st.has_prec) {
      for (uint32_t i = 0; i <= st.L; ++i) {
        st.pexp[i] = k.precinct_exps[i];
        if (i && ((st.pexp[i] & 15) == 0 || (st.pexp[i] >> 4) == 0)) return fail("precinct size too small");
      }
      for (uint32_t i = st.L + 1; i < 36; ++i) k.precinct_exps[i] = 0;
    } else memset(k.precinct_exps, 0, sizeof(k.precinct_exps));
  }
  for (uint32_t c = 0; c < p.num_comps; ++c) plan.max_decomps = std::max(plan.max_decomps, plan.style(c).L);
  if (p.color_transform && (plan.style(0).rev != plan.style(1).rev || plan.style(1).rev != plan.style(2).rev))
    return fail("When the colour transform is employed, all colour components must undergo either reversible or irreversible wavelet transform");   // ojph_tile.cpp:147-163
  {                                                   // ojph_codestream_local.cpp:582-620, ojph_tile.cpp:225-250
    uint32_t div = p.reserved[1] & 3u;
    if ((p.prog_order == 0 || p.prog_order == 1) && div == 2) div |= 1;     // LRCP / RLCP: per component means per resolution and component
    if (p.prog_order == 2) div &= ~2u;                                      // RPCL: only per resolution
    if (p.prog_order == 3) div = 0;                                         // PCRL: none
    if (p.prog_order == 4) div &= ~1u;                                      // CPRL: only per component
    p.reserved[1] = div;
    plan.tilepart_div = div;
    // RC divisions number the tile-parts c + r * num_comps and skip the (c, r) a component with fewer
    // decompositions does not have (ojph_tile.cpp:637-652); the 255 limit counts the ones that exist (:84-103)
    const uint32_t ML = plan.max_decomps;
    plan.parts_per_tile = div == 0 ? 1 : div == 1 ? ML + 1 : div == 2 ? p.num_comps : p.num_comps * (ML + 1);
    uint32_t existing = plan.parts_per_tile;
    if (div == 3) { existing = 0; for (uint32_t c = 0; c < p.num_comps; ++c) existing += plan.style(c).L + 1; }
    if (existing > 255) return fail("a tile cannot have more than 255 tile parts");
    if (plan.parts_per_tile > 255) return fail("tile-part numbers beyond 255 (components with fewer decompositions leave gaps)");
  }
  plan.p = p;
  // (a parsed codestream brings its own QCD / QCC: what the writer could not derive for these parameters -- more than 38 bits,
  // a kernel gain beyond the guard bits -- does not keep it from being read)
  if (!derive_quant(plan)) {
    if (!plan.parsed) return OJPHGPU_E_INVALID;
    plan.error.clear();                                            // place holders until the parser puts the codestream's own in
    plan.qcc.assign(p.num_comps, QuantSet()); plan.qcc_order.clear();
    plan.qcd = QuantSet(); plan.qcd.sqcd = 1u << 5; plan.qcd.guard_bits = 1;
    plan.qcd.q8.assign(97, (uint8_t)(8u << 3)); plan.qcd.q16.assign(97, (uint16_t)(8u << 11));
  }
  const bool parsed_nlt = p.nlt_bd_default != 0 || std::any_of(p.nlt_bd, p.nlt_bd + OJPHGPU_MAX_COC_COMPS, [](uint8_t v) { return v != 0; }) || p.nlt_reserved[0] != 0;
  if (!derive_nlt(plan, parsed_nlt)) return OJPHGPU_E_INVALID;

  plan.ntx = div_ceil(X1 - p.tile_x0, p.tile_w);                                   // ojph_codestream_local.cpp:113-123
  plan.nty = div_ceil(Y1 - p.tile_y0, p.tile_h);
  if ((uint64_t)plan.ntx * plan.nty > 65535) return fail("the number of tiles cannot exceed 65535");
  {
    // Host resources are bounded BEFORE the tables are built (a 60-byte header can announce 2^36 samples in 4x4
    // code-blocks): an upper estimate of the code-blocks -- every component's samples over its smallest possible
    // block (a precinct of 2^PP halves it once more per axis), 4/3 for the pyramid, plus the partial blocks at the
    // edges of every band of every tile-component -- must stay below 2^24 (about 1.5 GB of host tables).
    double est = 0;
    const double tiles = (double)plan.ntx * plan.nty;
    for (uint32_t c = 0; c < p.num_comps; ++c) {
      const CodStyle& st = plan.style(c);
      uint32_t lbw = st.lbw, lbh = st.lbh;
      for (uint32_t r = 0; r <= st.L && r < 36; ++r) { lbw = std::min(lbw, std::max(st.lpw(r), 1u) - (r ? 1u : 0u)); lbh = std::min(lbh, std::max(st.lph(r), 1u) - (r ? 1u : 0u)); }
      const double area = (double)(1u << lbw) * (double)(1u << lbh);
      est += (double)plan.comps[c].w * plan.comps[c].h / area * 1.34 + tiles * (3.0 * st.L + 1.0) * 4.0;
      est += tiles * ((double)plan.comps[c].w / std::max(plan.ntx, 1u) / (1u << lbw) + (double)plan.comps[c].h / std::max(plan.nty, 1u) / (1u << lbh)) * 2.0 * (3.0 * st.L + 1.0);
    }
    if (est > (double)(1u << 24)) return fail("too many code-blocks for the host tables (more than 2^24)");
  }
  for (uint32_t ty = 0; ty < plan.nty; ++ty)
    for (uint32_t tx = 0; tx < plan.ntx; ++tx) {
      Tile t; t.idx = ty * plan.ntx + tx;
      {                                                                           // ojph_codestream_local.cpp:133-163
        const uint64_t gx0 = (uint64_t)p.tile_x0 + (uint64_t)tx * p.tile_w, gy0 = (uint64_t)p.tile_y0 + (uint64_t)ty * p.tile_h;
        t.r.x0 = (uint32_t)std::max<uint64_t>(gx0, p.image_x0); t.r.y0 = (uint32_t)std::max<uint64_t>(gy0, p.image_y0);
        t.r.w = (uint32_t)std::min<uint64_t>(gx0 + p.tile_w, X1) - t.r.x0;
        t.r.h = (uint32_t)std::min<uint64_t>(gy0 + p.tile_h, Y1) - t.r.y0;
      }
      for (uint32_t c = 0; c < p.num_comps; ++c) {
        const CompGeo& cg = plan.comps[c];
        const CodStyle& st = plan.style(c);
        const uint32_t L = st.L, lbw = st.lbw, lbh = st.lbh;
        TileComp tc; tc.tile = t.idx; tc.comp = c;
        tc.r.x0 = div_ceil(t.r.x0, cg.dx); tc.r.y0 = div_ceil(t.r.y0, cg.dy);      // ojph_tile.cpp:262-275
        tc.r.w = div_ceil(t.r.x0 + t.r.w, cg.dx) - tc.r.x0; tc.r.h = div_ceil(t.r.y0 + t.r.h, cg.dy) - tc.r.y0;
        tc.res.assign(L + 1, 0);
        // resolution rectangles, top down (ojph_resolution.cpp:302-330 with band 0)
        // (a DFS marker segment may have a level halve one direction only, or none: ojph_resolution.cpp:302-395)
        std::vector<Rect> rr(L + 1);
        std::vector<uint32_t> kinds(L + 1, 1), dsx(L + 1, 1), dsy(L + 1, 1);   // kind of the level that splits r; down-sampling of r against the component
        rr[L] = tc.r;
        for (uint32_t r = L; r > 0; --r) {
          const uint32_t kind = plan.level_kind(c, L - r + 1);
          const bool hx = kind == 1 || kind == 2, hy = kind == 1 || kind == 3;
          kinds[r] = kind;
          Rect a = rr[r], b = a;
          if (hx) { b.x0 = (a.x0 + 1) >> 1; b.w = ((a.x0 + a.w + 1) >> 1) - b.x0; }
          if (hy) { b.y0 = (a.y0 + 1) >> 1; b.h = ((a.y0 + a.h + 1) >> 1) - b.y0; }
          rr[r - 1] = b;
          dsx[r - 1] = dsx[r] * (hx ? 2u : 1u); dsy[r - 1] = dsy[r] * (hy ? 2u : 1u);
        }
        for (uint32_t r = 0; r <= L; ++r) {
          Resolution R; R.tile = t.idx; R.comp = c; R.res = r; R.r = rr[r]; R.kind = r ? kinds[r] : 1;
          const bool hx = R.kind == 1 || R.kind == 2, hy = R.kind == 1 || R.kind == 3;
          const uint32_t lpw = st.lpw(r), lph = st.lph(r);       // this resolution's precinct size
          R.log_ppw = lpw; R.log_pph = lph;
          for (int i = 0; i < 4; ++i) R.band[i] = -1;
          R.plane_off = 0; R.pitch = 0;
          // (the raw plane of a resolution r > 0 -- res 0 lives in its LL band -- and the band planes get their place in
          // the arena from assign_planes, once the sample width of the component is known)
          uint32_t trx0 = R.r.x0, try0 = R.r.y0, trx1 = R.r.x0 + R.r.w, try1 = R.r.y0 + R.r.h;
          const uint32_t xoff = (r > 0 && hx) ? 1 : 0, yoff = (r > 0 && hy) ? 1 : 0;   // subband::finalize_alloc x_off / y_off (ojph_subband.cpp:134-138)
          for (uint32_t b = (r ? 1 : 0); b < (r ? 4u : 1u); ++b) {
            if (r > 0 && !(R.kind == 1 || (R.kind == 2 && b == 1) || (R.kind == 3 && b == 2))) continue;   // the bands this kind of level has
            Band B; B.tile = t.idx; B.comp = c; B.res = r; B.band = b;
            if (r > 0) {
              B.r = R.r;
              if (hx) { B.r.x0 = (trx0 - (b & 1) + 1) >> 1; B.r.w = ((trx1 - (b & 1) + 1) >> 1) - B.r.x0; }
              if (hy) { B.r.y0 = (try0 - (b >> 1) + 1) >> 1; B.r.h = ((try1 - (b >> 1) + 1) >> 1) - B.r.y0; }
            } else B.r = R.r;
            B.K_max = band_Kmax(plan, c, r, b);
            B.delta = 0.0f; B.delta_inv = 0.0f;
            if (!st.rev) {                                         // ojph_subband.cpp:156-164
              float d = band_delta(plan, c, r, b);
              d /= (float)(1u << ((31u - B.K_max) & 31u));          // (K_max beyond 31: refused further on -- writer: make_quant, parser: its own K_max test)
              B.delta = d; B.delta_inv = 1.0f / d;
            }
            B.xcb = std::min(lbw, lpw - xoff); B.ycb = std::min(lbh, lph - yoff);
            B.empty = (B.r.w == 0 || B.r.h == 0);
            B.nbx = B.nby = 0; B.first_block = (uint32_t)plan.blocks.size();
            B.plane_off = 0; B.pitch = 0;
            if (!B.empty) {
              uint32_t x1 = B.r.x0 + B.r.w, y1 = B.r.y0 + B.r.h;
              B.nbx = ((x1 + (1u << B.xcb) - 1) >> B.xcb) - (B.r.x0 >> B.xcb);
              B.nby = ((y1 + (1u << B.ycb) - 1) >> B.ycb) - (B.r.y0 >> B.ycb);
              uint32_t xl = (B.r.x0 >> B.xcb) << B.xcb, yl = (B.r.y0 >> B.ycb) << B.ycb;
              for (uint32_t by = 0; by < B.nby; ++by)
                for (uint32_t bx = 0; bx < B.nbx; ++bx) {
                  Block k; k.band = (uint32_t)plan.bands.size(); k.bx = bx; k.by = by;
                  uint32_t cx0 = std::max(B.r.x0, xl + (bx << B.xcb));
                  uint32_t cx1 = std::min(x1, xl + ((bx + 1) << B.xcb));
                  uint32_t cy0 = std::max(B.r.y0, yl + (by << B.ycb));
                  uint32_t cy1 = std::min(y1, yl + ((by + 1) << B.ycb));
                  k.r.x0 = cx0 - B.r.x0; k.r.y0 = cy0 - B.r.y0; k.r.w = cx1 - cx0; k.r.h = cy1 - cy0;
                  plan.blocks.push_back(k);
                  if (plan.blocks.size() > 0x3FFFFFFFull) return fail("too many code-blocks");
                  plan.max_block_bytes = std::max(plan.max_block_bytes,
                                                  block_scratch_bytes(k.r.w, k.r.h, B.K_max));
                }
            }
            R.band[b] = (int)plan.bands.size();
            plan.bands.push_back(B);
          }
          // precincts (ojph_resolution.cpp:401-441)
          R.npw = R.nph = 0; R.first_precinct = (uint32_t)plan.precincts.size();
          if (trx0 != trx1 && try0 != try1) {
            R.npw = ((trx1 + (1u << lpw) - 1) >> lpw) - (trx0 >> lpw);
            R.nph = ((try1 + (1u << lph) - 1) >> lph) - (try0 >> lph);
            uint32_t xlb = (trx0 >> lpw) << lpw, ylb = (try0 >> lph) << lph;
            for (uint32_t y = 0; y < R.nph; ++y)
              for (uint32_t x = 0; x < R.npw; ++x) {
                Precinct P; P.tile = t.idx; P.comp = c; P.res = r;
                uint64_t ix = (uint64_t)dsx[r] * cg.dx * (xlb + (x << lpw)), iy = (uint64_t)dsy[r] * cg.dy * (ylb + (y << lph));
                P.img_x = (uint32_t)std::max<uint64_t>(ix, t.r.x0);
                P.img_y = (uint32_t)std::max<uint64_t>(iy, t.r.y0);
                for (int i = 0; i < 4; ++i) P.cb[i] = Rect{0, 0, 0, 0};
                plan.precincts.push_back(P);
              }
            // code-block index rectangles per band (ojph_subband.cpp:224-276)
            for (uint32_t b = 0; b < 4; ++b) {
              if (R.band[b] < 0) continue;
              const Band& B = plan.bands[(size_t)R.band[b]];
              if (B.empty) continue;
              uint32_t pc_l = (trx0 >> lpw) << lpw, pc_t = (try0 >> lph) << lph;
              uint32_t xs = xoff, ys = yoff, coly = 0;       // subband::get_cb_indices x_shift / y_shift (ojph_subband.cpp:240-241)
              for (uint32_t y = 0; y < R.nph; ++y) {
                uint32_t pcy0 = std::max(try0, pc_t + (y << lph));
                uint32_t pcy1 = std::min(try1, pc_t + ((y + 1) << lph));
                pcy0 = (pcy0 - (b >> 1) + (1u << ys) - 1) >> ys;
                pcy1 = (pcy1 - (b >> 1) + (1u << ys) - 1) >> ys;
                uint32_t yb = ((pcy1 + (1u << B.ycb) - 1) >> B.ycb) - (pcy0 >> B.ycb);
                uint32_t colx = 0;
                for (uint32_t x = 0; x < R.npw; ++x) {
                  uint32_t pcx0 = std::max(trx0, pc_l + (x << lpw));
                  uint32_t pcx1 = std::min(trx1, pc_l + ((x + 1) << lpw));
                  pcx0 = (pcx0 - (b & 1) + (1u << xs) - 1) >> xs;
                  pcx1 = (pcx1 - (b & 1) + (1u << xs) - 1) >> xs;
                  uint32_t xb = ((pcx1 + (1u << B.xcb) - 1) >> B.xcb) - (pcx0 >> B.xcb);
                  Precinct& P = plan.precincts[R.first_precinct + y * R.npw + x];
                  P.cb[b] = Rect{colx, coly, xb, yb};
                  colx += xb;
                }
                coly += yb;
              }
            }
          }
          tc.res[r] = (uint32_t)plan.ress.size();
          plan.ress.push_back(R);
        }
        t.comps.push_back((uint32_t)plan.tcomps.size());
        plan.tcomps.push_back(tc);
      }
      plan.tiles.push_back(t);
    }
  if (!derive_precision(plan)) return OJPHGPU_E_INVALID;
  assign_planes(plan);

  // packet (precinct) order per tile (ojph_tile.cpp:604-772)
  for (Tile& t : plan.tiles) {
    auto res_of = [&](uint32_t c, uint32_t r) -> const Resolution& {
      return plan.ress[plan.tcomps[t.comps[c]].res[r]];
    };
    // a component with fewer decompositions than the largest simply has no resolution r > its own
    // (tile_comp::write_precincts / get_top_left_precinct find nothing left to write, ojph_tile_comp.cpp:116-160)
    const uint32_t nc = p.num_comps, L = plan.max_decomps;
    auto has = [&](uint32_t c, uint32_t r) { return r <= plan.style(c).L; };
    std::vector<std::vector<uint32_t>> cur(nc, std::vector<uint32_t>(L + 1, 0));
    auto top = [&](uint32_t c, uint32_t r, uint32_t& x, uint32_t& y) {
      if (!has(c, r)) return false;
      const Resolution& R = res_of(c, r);
      if (cur[c][r] >= R.npw * R.nph) return false;
      const Precinct& P = plan.precincts[R.first_precinct + cur[c][r]];
      x = P.img_x; y = P.img_y; return true;
    };
    auto emit = [&](uint32_t c, uint32_t r) {
      const Resolution& R = res_of(c, r);
      t.packets.push_back(R.first_precinct + cur[c][r]); cur[c][r]++;
    };
    if (p.prog_order == 0 || p.prog_order == 1) {
      for (uint32_t r = 0; r <= L; ++r)
        for (uint32_t c = 0; c < nc; ++c) {
          if (!has(c, r)) continue;
          const Resolution& R = res_of(c, r);
          for (uint32_t i = 0; i < R.npw * R.nph; ++i) emit(c, r);
        }
    } else if (p.prog_order == 2) {
      for (uint32_t r = 0; r <= L; ++r)
        for (;;) {
          bool found = false; uint32_t bc = 0, sx = 0x7FFFFFFF, sy = 0x7FFFFFFF, x, y;
          for (uint32_t c = 0; c < nc; ++c) {
            if (!top(c, r, x, y)) continue;
            found = true;
            if (y < sy || (y == sy && x < sx)) { sx = x; sy = y; bc = c; }
          }
          if (!found) break;
          emit(bc, r);
        }
    } else if (p.prog_order == 3) {
      for (;;) {
        bool found = false; uint32_t bc = 0, br = 0, sx = 0x7FFFFFFF, sy = 0x7FFFFFFF, x, y;
        for (uint32_t c = 0; c < nc; ++c)
          for (uint32_t r = 0; r <= L; ++r) {
            if (!top(c, r, x, y)) continue;
            found = true;
            if (y < sy || (y == sy && x < sx) || (y == sy && x == sx && c < bc) ||
                (y == sy && x == sx && c == bc && r < br)) { sx = x; sy = y; bc = c; br = r; }
          }
        if (!found) break;
        emit(bc, br);
      }
    } else {
      for (uint32_t c = 0; c < nc; ++c)
        for (;;) {
          bool found = false; uint32_t br = 0, sx = 0x7FFFFFFF, sy = 0x7FFFFFFF, x, y;
          for (uint32_t r = 0; r <= L; ++r) {
            if (!top(c, r, x, y)) continue;
            found = true;
            if (y < sy || (y == sy && x < sx)) { sx = x; sy = y; br = r; }
          }
          if (!found) break;
          emit(c, br);
        }
    }
  }
  return OJPHGPU_OK;
}

}  // namespace ojphgpu

// ---------------------------------------------------------------------------------------------
// C ABI: plan
// ---------------------------------------------------------------------------------------------
using namespace ojphgpu;

extern "C" int ojphgpu_plan_create(const ojphgpu_params* params, ojphgpu_plan** out)
{
  if (!params || !out) return OJPHGPU_E_INVALID;
  *out = nullptr;
  ojphgpu_plan* h = new (std::nothrow) ojphgpu_plan();
  if (!h) return OJPHGPU_E_NOMEM;
  const int rc = no_throw([&] { return build_plan(*params, h->plan); });
  if (rc != OJPHGPU_OK) { delete h; return rc; }
  *out = h;
  return OJPHGPU_OK;
}

extern "C" void ojphgpu_plan_destroy(ojphgpu_plan* plan) { delete plan; }

extern "C" int ojphgpu_plan_params(const ojphgpu_plan* plan, ojphgpu_params* out)
{
  if (!plan || !out) return OJPHGPU_E_INVALID;
  *out = plan->plan.p;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_counts(const ojphgpu_plan* plan, uint64_t out[8])
{
  if (!plan || !out) return OJPHGPU_E_INVALID;
  const Plan& P = plan->plan;
  out[0] = P.tiles.size(); out[1] = P.bands.size(); out[2] = P.blocks.size();
  out[3] = P.levels.size(); out[4] = P.arena_elems; out[5] = P.max_block_bytes;
  out[6] = P.precincts.size(); out[7] = P.tcomps.size();
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_set_comments(ojphgpu_plan* plan, const uint8_t* const* data, const uint16_t* len,
                                          const uint16_t* rcom, uint32_t n)
{
  if (!plan || (n && (!data || !len || !rcom))) return OJPHGPU_E_INVALID;
  return no_throw([&] {
    std::vector<Plan::Comment> c(n);
    for (uint32_t i = 0; i < n; ++i) {
      if (len[i] > 65531 || (len[i] && !data[i])) return (int)OJPHGPU_E_INVALID;    // Lcom = len + 4 is 16 bits
      c[i].rcom = rcom[i]; c[i].data.assign(data[i], data[i] + len[i]);
    }
    plan->plan.comments.swap(c);
    return (int)OJPHGPU_OK;
  });
}

extern "C" int ojphgpu_plan_tile_parts(const ojphgpu_plan* plan, uint32_t* parts_per_tile)
{
  if (!plan || !parts_per_tile) return OJPHGPU_E_INVALID;
  *parts_per_tile = plan->plan.parts_per_tile;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_bands(const ojphgpu_plan* plan, ojphgpu_band_info* out, size_t n)
{
  if (!plan || !out || n < plan->plan.bands.size()) return OJPHGPU_E_INVALID;
  size_t i = 0;
  for (const Band& B : plan->plan.bands) {
    ojphgpu_band_info& o = out[i++];
    o.tile = B.tile; o.comp = B.comp; o.res = B.res; o.band = B.band;
    o.x0 = B.r.x0; o.y0 = B.r.y0; o.w = B.r.w; o.h = B.r.h; o.K_max = B.K_max;
    o.delta = B.delta; o.delta_inv = B.delta_inv; o.nbx = B.nbx; o.nby = B.nby;
    o.first_block = B.first_block; o.plane_off = B.plane_off; o.pitch = B.pitch; o.reserved = 0;
  }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_blocks(const ojphgpu_plan* plan, ojphgpu_block_info* out, size_t n)
{
  if (!plan || !out || n < plan->plan.blocks.size()) return OJPHGPU_E_INVALID;
  size_t i = 0;
  for (const Block& k : plan->plan.blocks) {
    ojphgpu_block_info& o = out[i++];
    o.band = k.band; o.x0 = k.r.x0; o.y0 = k.r.y0; o.w = k.r.w; o.h = k.r.h;
    o.K_max = plan->plan.bands[k.band].K_max;
  }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_levels(const ojphgpu_plan* plan, ojphgpu_level_info* out, size_t n)
{
  if (!plan || !out || n < plan->plan.levels.size()) return OJPHGPU_E_INVALID;
  if (!plan->plan.levels.empty())
    memcpy(out, plan->plan.levels.data(), sizeof(ojphgpu_level_info) * plan->plan.levels.size());
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_comp_plane(const ojphgpu_plan* plan, uint32_t tile, uint32_t comp,
                                        uint64_t* off, uint32_t* pitch, uint32_t rect[4])
{
  if (!plan) return OJPHGPU_E_INVALID;
  const Plan& P = plan->plan;
  if (tile >= P.tiles.size() || comp >= P.p.num_comps) return OJPHGPU_E_INVALID;
  const TileComp& tc = P.tcomps[P.tiles[tile].comps[comp]];
  uint32_t L = P.recon_decomps(comp);                       // reduced-resolution decoding reconstructs a lower resolution
  const Resolution& R = P.ress[tc.res[L]];
  if (L == 0) {
    const Band& B = P.bands[(size_t)R.band[0]];
    if (off) *off = B.plane_off;
    if (pitch) *pitch = B.pitch;
  } else {
    if (off) *off = R.plane_off;
    if (pitch) *pitch = R.pitch;
  }
  if (rect) { rect[0] = R.r.x0; rect[1] = R.r.y0; rect[2] = R.r.w; rect[3] = R.r.h; }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_comp_format(const ojphgpu_plan* plan, uint32_t comp, uint32_t* bit_depth, uint32_t* is_signed)
{
  if (!plan || comp >= plan->plan.p.num_comps) return OJPHGPU_E_INVALID;
  if (bit_depth) *bit_depth = plan->plan.comps[comp].bit_depth;
  if (is_signed) *is_signed = plan->plan.comps[comp].is_signed ? 1 : 0;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_comp_style(const ojphgpu_plan* plan, uint32_t comp, uint32_t out[8])
{
  if (!plan || !out || comp >= plan->plan.p.num_comps) return OJPHGPU_E_INVALID;
  const CodStyle& st = plan->plan.style(comp);
  memset(out, 0, 8 * sizeof(uint32_t));
  out[0] = st.L; out[1] = st.rev ? 1 : 0; out[2] = st.lbw; out[3] = st.lbh; out[4] = st.rank ? 1 : 0;
  out[5] = plan->plan.recon_decomps(comp);
  out[6] = plan->plan.nlt3[comp];
  out[7] = ((comp < plan->plan.wide.size() && plan->plan.wide[comp]) ? 1u : 0u)   // bit 0: the component takes the 64-bit sample path
         | (plan->plan.general(comp) ? 2u : 0u);                                  // bit 1: ... the general lifting kernels (that, or Part 2)
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_comp_lift(const ojphgpu_plan* plan, uint32_t comp, uint32_t level, ojphgpu_lift* out)
{
  if (!plan || !out || comp >= plan->plan.p.num_comps || level == 0) return OJPHGPU_E_INVALID;
  *out = plan->plan.lift_of(comp, level);
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_restrict_resolution(ojphgpu_plan* plan, uint32_t skipped_res_for_data, uint32_t skipped_res_for_recon)
{
  if (!plan) return OJPHGPU_E_INVALID;
  Plan& P = plan->plan;
  if (P.coded.size() != P.blocks.size()) return OJPHGPU_E_INVALID;           // a parsed codestream only
  if (skipped_res_for_data < skipped_res_for_recon) return OJPHGPU_E_INVALID; // ojph_codestream_local.cpp:886-890
  if (skipped_res_for_data > P.p.num_decomps) return OJPHGPU_E_INVALID;       // :891-895 (the COD's count)
  for (uint32_t c = 0; c < P.p.num_comps; ++c)                                // a component with fewer decompositions than are
    if (skipped_res_for_data > P.style(c).L) return OJPHGPU_E_INVALID;        // skipped: the reference's arithmetic wraps there
  P.skip_read = skipped_res_for_data; P.skip_recon = skipped_res_for_recon;
  // the reconstructed components: sub-sampling grows by 2^skip_recon (ojph_params.cpp:930-946)
  const uint64_t X1 = (uint64_t)P.p.image_x0 + P.p.width, Y1 = (uint64_t)P.p.image_y0 + P.p.height;
  P.frame_elems = 0;
  uint32_t ci = 0;
  for (CompGeo& g : P.comps) {
    // (with a DFS marker segment a skipped level may halve one direction only: param_dfs::get_res_downsamp, :2575-2593)
    uint64_t fx = g.dx, fy = g.dy;
    for (uint32_t d = 1; d <= P.skip_recon; ++d) {
      const uint32_t kind = P.level_kind(ci, d);
      if (kind == 1 || kind == 2) fx *= 2;
      if (kind == 1 || kind == 3) fy *= 2;
    }
    ++ci;
    g.x0 = (uint32_t)((P.p.image_x0 + fx - 1) / fx); g.y0 = (uint32_t)((P.p.image_y0 + fy - 1) / fy);
    g.w = (uint32_t)((X1 + fx - 1) / fx) - g.x0; g.h = (uint32_t)((Y1 + fy - 1) / fy) - g.y0;
    g.frame_off = P.frame_elems;
    P.frame_elems += (uint64_t)g.w * g.h;
  }
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_comp_info(const ojphgpu_plan* plan, uint32_t comp, uint32_t out[8])
{
  if (!plan || !out) return OJPHGPU_E_INVALID;
  const Plan& P = plan->plan;
  if (comp > P.p.num_comps) return OJPHGPU_E_INVALID;
  memset(out, 0, 8 * sizeof(uint32_t));
  if (comp == P.p.num_comps) { out[4] = (uint32_t)P.frame_elems; out[5] = (uint32_t)(P.frame_elems >> 32); return OJPHGPU_OK; }
  const CompGeo& g = P.comps[comp];
  out[0] = g.x0; out[1] = g.y0; out[2] = g.w; out[3] = g.h;
  out[4] = (uint32_t)g.frame_off; out[5] = (uint32_t)(g.frame_off >> 32); out[6] = g.dx; out[7] = g.dy;
  return OJPHGPU_OK;
}

extern "C" int ojphgpu_plan_coded_blocks(const ojphgpu_plan* plan, ojphgpu_coded_block* out, size_t n)
{
  if (!plan || !out) return OJPHGPU_E_INVALID;
  const Plan& P = plan->plan;
  if (P.coded.size() != P.blocks.size() || n < P.coded.size()) return OJPHGPU_E_INVALID;
  for (size_t i = 0; i < P.coded.size(); ++i) {
    out[i].offset = P.coded[i].offset; out[i].len1 = P.coded[i].len1; out[i].len2 = P.coded[i].len2;
    out[i].missing_msbs = P.coded[i].missing_msbs; out[i].num_passes = P.coded[i].num_passes;
  }
  return OJPHGPU_OK;
}

extern "C" const char* ojphgpu_version(void) { return "openjph_amd 0.1 (HTJ2K hot path for gfx950; compatible with OpenJPH 0.31.0)"; }
