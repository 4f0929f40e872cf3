// pik_exact.hpp -- gradient descent of the EXACT flavour (PIK_STRICT: bit-identical to the CPU oracle)
// with the accept evaluation's work re-used by the 2D finite-difference probes.
//
// step() of the reference (src/ik_gradient.cpp:28-43) perturbs ONE variable per probe:
//     cost_fn(local - h e_i), cost_fn(local + h e_i),   i = 0 .. D-1
// and the live forward kinematics multiplies the chain left to right (src/fk_moveit.cpp:20-33 ->
// RobotState::updateLinkTransforms; oracle/pik_oracle.c fk_path):
//     g <- g * origin_j;  g <- g * joint_j(q_j)          j = 0 .. D-1,   then  g <- g * tip.
// A probe of variable i therefore repeats, operation for operation on the same inputs, everything the
// evaluation of `local` itself did up to and including `g * origin_i`, and the sine / cosine of every
// other joint.  Floating-point arithmetic is deterministic: the repeated operations give the repeated
// bits.  So the probe is computed as
//     (the accept evaluation's frame in front of joint i) * joint_i(q_i +- h) * origin_{i+1} *
//     joint_{i+1}(sin, cos of the accept evaluation) * ... * tip,        then the pose / joint costs,
// i.e. ONE sincos and the products from joint i to the tip instead of D sincos and the whole chain -- the
// same numbers as the literal evaluation (the products behind joint i are NOT shared: they multiply a
// different left factor, and the oracle's product is not associative).  The reference's evaluation
// COUNTER stays literal (2D + 3 per step).
//
// Two forms:
//   * LPE = 1, 2 lanes per elite ("fork"): the probes of joint i branch off the accept evaluation's walk
//     down the chain at the moment it stands in front of joint i -- no frame is stored anywhere; with two
//     lanes the pair takes the - h and the + h probe side by side.  The accept evaluation of step k and the
//     probes of step k + 1 are the same point, so they are one walk (skipped when no lane can take another
//     step: the iteration limit is known beforehand).
//   * LPE >= 4 ("passes"): the lanes of an elite hold the same state and all repeat the accept evaluation;
//     lane 0 leaves the 12 D numbers of the frames in front of the joints in LDS (once per ELITE, not per
//     lane), and the 2D probes are dealt out LPE at a time as before, every lane starting from the frame of
//     ITS joint; a pass walks the joints from its first probe's joint to the tip in lock-step.
//
// Floating joints and several tip frames keep the literal routine (gradient_descent, PIK_STRICT paths).
#pragma once

#if defined(PIK_STRICT)

#ifndef PIK_EXACT_PAIRED
#define PIK_EXACT_PAIRED 1
#endif

// The team evaluations and probe passes (4..16 lanes per elite) are INLINED into the descent for chains of up to
// PIK_XTEAM_INLINE_MAXD variables (eight until round 6; nine and ten since: the Panda on a two-joint torso 8.72 -> 6.82 ms
// per 4096-target step with the forms of class 2 and everything inlined, 7.75 with the forms alone; the 2- and 4-lane
// kernels of ten variables then spill 130-190 vector registers -- fuzz suite and self test green) and real calls beyond: a call costs the callee's saves of every callee-saved
// register it touches, the joint vector through memory, and a wait on each -- on the critical path of the lone
// wavefronts that run these variants (interleaved A/B on one box, seven variables: 59.65 -> 58.1 ms on the driver's
// pool).  The long chains keep the calls: their descent sits at the register cap as it is.
// The one-lane fork stays a call at every length (inlined: 59.3 ms, within the noise of 59.65).
#ifndef PIK_XTEAM_INLINE_MAXD
#define PIK_XTEAM_INLINE_MAXD 10
#endif
#ifndef PIK_XGD_REGS_OCC2
#define PIK_XGD_REGS_OCC2 0
#endif
// The one-lane fork and the line-search pair likewise, up to PIK_XFORK_INLINE_MAXD variables: each is a function of
// ~248 vector registers, whose prologue saves every callee-saved register it touches -- about a hundred, 400 bytes
// per lane in and out, twice per descent step.  In time that was nearly free (59.65 -> 59.3 ms: two wavefronts per
// SIMD hide it, a lone one has AGPRs), but it was the exact kernels' HBM traffic: ~500 KB per solved problem against
// 180 bytes of algorithmic I/O (profiles/r05base_driver_cmd_exact_summary.txt: WRITE_SIZE of
// memetic_kernel<7,1,false,2>).  Inlined, the descent saves its registers once per generation.
#ifndef PIK_XPAIR_OCC2
#define PIK_XPAIR_OCC2 1 // (experiment, 0: the two-per-SIMD kernels walk their probes one at a time -- fewer registers)
#endif
#ifndef PIK_XFORK_INLINE_MAXD
#define PIK_XFORK_INLINE_MAXD 10
#endif
// The one-lane descent's point and gradient in the lane's LDS column (ExactLds::Q0 / G0); 0: as arrays (A/B experiments)
// (chains of up to PIK_XFORK_INLINE_MAXD variables, whose evaluations are inlined: with the long chains' calls the
//  copies of the point around them cost registers the kernels at the cap do not have -- 240-250 more spilled ones)
#ifndef PIK_XLDS_STATE
#define PIK_XLDS_STATE 1
#endif
// FLAT probe passes (chains of class 1 / 2, 4..16 lanes per elite): every lane of a pass starts from the SAME frame --
// the accept evaluation's frame in front of the pass's first joint, or nothing at all when one pass holds every probe --
// and walks the SAME joints, with its own sine / cosine at its own joint and the accept evaluation's everywhere else.
// In front of its joint a lane thereby repeats the accept evaluation's operations on the accept evaluation's operands
// (the same bits as the stored frame it used to start from), and no instruction of the walk is predicated: the
// per-lane start (`if (j > i)`, `if (j >= i)`) cost, per joint, a copy of the whole frame (16 + 12 register moves to
// merge the lanes that had run the block with those that had not) and the exec-mask bookkeeping around it -- as many
// instructions as the joint's arithmetic.  0: the per-lane start (A/B experiments).
#ifndef PIK_XFLAT
#define PIK_XFLAT 1
#endif
// ... for chains of up to PIK_XFLAT_MAXD variables: with eight, the unrolled walks of the 4- and 8-lane descents -- at
// the register cap as they are -- spill 330-470 vector registers (kernel resource ledger)
#ifndef PIK_XFLAT_MAXD
#define PIK_XFLAT_MAXD 7
#endif

namespace pik {

// The called evaluations get their LDS blocks as LDS pointers (address space 3): as generic pointers every access
// was a FLAT instruction -- the slow path to LDS, and one more kind of wait on a lone wavefront's critical path.
typedef __attribute__((address_space(3))) double LdsF64;

// what a called evaluation hands back, in registers (an EvalOut& is a round trip through the stack)
struct CostSol {
    double cost;
    int sol;
};
// (the joint vector stays an array reference: by value -- registers for up to eight variables -- measured slower,
//  81.1 against 79.8 ms on the driver's pool, interleaved on one box: the copies around the calls cost more than the
//  load they save)

// ---- UZ: the chain class "every variable a revolute joint about its frame's +z, no identity origin, a tip
// transform, no mimic joint" (ChainK::uniform_z, decided on the host: Franka Panda, KUKA iiwa, any description
// written in the Denavit-Hartenberg convention).  What the general routines decide per joint at run time -- origin
// skipped?  prismatic?  which axis? -- is a compile-time constant here, the joint loops of the team evaluations are
// unrolled (constant addresses), and the arithmetic of the path taken is the same, operation for operation
// (chain_origin / rotate_exact AXIS_Z / iso_mul): the same bits.
template <int D, int UZ>
__device__ __forceinline__ void x_origin(CK<D> c, int j, double (&R)[9], double (&t)[3], bool blank) {
    if constexpr (UZ) {
        CPtr o = c.O[j];
        if (blank) {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = o[i];
            t[0] = o[9];
            t[1] = o[10];
            t[2] = o[11];
        } else {
            x_iso_mul<UZ>(R, t, o, PIK_OKIND(c, j), PIK_OPM(c, j));
        }
    } else {
        chain_origin<D>(c, j, R, t, blank);
    }
}
// two frames by the origin of joint j (never blank)
template <int D, int UZ>
__device__ __forceinline__ void x_origin_pair(CK<D> c, int j, double (&Ra)[9], double (&ta)[3], double (&Rb)[9],
                                              double (&tb)[3]) {
    if constexpr (UZ) {
        x_iso_mul_pair<UZ>(Ra, ta, Rb, tb, c.O[j], PIK_OKIND(c, j), PIK_OPM(c, j));
    } else {
        chain_origin<D>(c, j, Ra, ta, false);
        chain_origin<D>(c, j, Rb, tb, false);
    }
}
template <int D, int UZ>
__device__ __forceinline__ void x_joint(CK<D> c, int j, double (&R)[9], double (&t)[3], bool prismatic, uint32_t kind,
                                        double v, double sn, double cs) {
    if constexpr (UZ) {
        (void)prismatic;
        (void)v;
        x_rotate<UZ>(R, kind, sn, cs);
    } else {
        chain_joint<D>(c, j, R, t, prismatic, kind, v, sn, cs);
    }
}
// two frames through joint j (values va / vb)
template <int D, int UZ>
__device__ __forceinline__ void x_joint_pair(CK<D> c, int j, double (&Ra)[9], double (&ta)[3], double (&Rb)[9],
                                             double (&tb)[3], bool prismatic, uint32_t kind, double va, double sna,
                                             double csa, double vb, double snb, double csb) {
    if constexpr (UZ) {
        (void)prismatic;
        (void)va;
        (void)vb;
        (void)ta;
        (void)tb;
        x_rotate_pair<UZ>(Ra, Rb, kind, sna, csa, snb, csb);
    } else {
        chain_joint<D>(c, j, Ra, ta, prismatic, kind, va, sna, csa);
        chain_joint<D>(c, j, Rb, tb, prismatic, kind, vb, snb, csb);
    }
}
template <int D, int UZ>
__device__ __forceinline__ void x_tip(CK<D> c, double (&R)[9], double (&t)[3]) {
    if constexpr (UZ) {
        x_tip_mul<UZ>(R, t, c.tip, c.tip_kind, PIK_TPM(c));
    } else {
        if (!c.tip_ident) iso_mul(R, t, c.tip);
    }
}
template <int D, int UZ>
__device__ __forceinline__ void x_tip_pair(CK<D> c, double (&Ra)[9], double (&ta)[3], double (&Rb)[9], double (&tb)[3]) {
    if constexpr (UZ) {
        x_tip_mul_pair<UZ>(Ra, ta, Rb, tb, c.tip, c.tip_kind, PIK_TPM(c));
    } else {
        if (!c.tip_ident) {
            iso_mul(Ra, ta, c.tip);
            iso_mul(Rb, tb, c.tip);
        }
    }
}

// LDS rows (64 doubles each; row r of lane l at [r * 64 + l]) of gradient_descent_exact
template <int D, int LPE>
struct ExactLds {
    // LPE <= 2: everything in the lane's own column
    static constexpr int SN0 = 0;      // [D] sine of every joint at the accepted point
    static constexpr int CS0 = D;      // [D] cosine
    // cost of the probes local -+ h e_i (LPE <= 2: the lane's own column; LPE >= 4: the column of the elite's
    // first lane)
    static constexpr int CM0 = LPE >= 4 ? 0 : 2 * D;
    static constexpr int CP0 = LPE >= 4 ? D : 3 * D;
    // LPE = 1 with the probes of a variable walked as a pair (L1P, the default): the lane's own column holds [sn D]
    // [cs D][gradient D][joint vector D] -- row G0 + i the DIFFERENCE cost(q + h e_i) - cost(q - h e_i) the fork forms
    // on the spot, then the normalised gradient; rows Q0 the accepted point itself.  The descent's point and gradient
    // are read from there wherever the joint is not a compile-time constant (the rolled walks), and where they
    // were read from the caller's frame (two wavefronts per SIMD: the state lives in memory) -- every such read was a
    // memory round trip with its wait in front of the arithmetic, ~20 per descent step.
    static constexpr int G0 = 2 * D;
    static constexpr int Q0 = 3 * D;
    // LPE >= 4: cost / verdict of the accept evaluation when it rides along with the probes (rows of 64, the column
    // of the elite's first lane)
    static constexpr int AC0 = 2 * D;
    static constexpr int AS0 = 2 * D + 1;
    // LPE >= 4: one block per ELITE (slot = lane / LPE): sines, cosines and values of the joints at the accepted
    // point, then the frames in front of the joints and the frame behind the last one -- [sn D][cs D][q D]
    // [frame 12 (D + 1)] -- then 12 more numbers.  The two line-search evaluations re-use it: the team of q - g the
    // first 3 D numbers, the team of q + g the next 3 D; the rows of a team's final frame are exchanged through
    // the last 24 numbers (12 per team; the first 12 are "frame D" of the accept evaluation before that).
    static constexpr int EB0 = 2 * D + 2;
    static constexpr int EBX = 15 * D;      // frame D / exchange slot of team 0
    static constexpr int EBS = 15 * D + 24;
    static constexpr int ROWS = LPE >= 4 ? EB0 + (EBS * (WAVE / LPE) + WAVE - 1) / WAVE : 4 * D;
};

// The accept evaluation of q (cost + verdict, exactly `evaluate`) at LPE <= 2, leaving every joint's sine /
// cosine in this lane's LDS column, and, when `want`, the costs of the probes q -+ h e_i in rows CM0 + i / CP0 + i
// (LPE = 2: this lane's sign only).
// (OCC: see evaluate)
template <int D, int LPE, int OCC = 1, int UZG = 0>
__device__ __forceinline__ void exact_accept_impl(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                          const double (&q_in)[D], EvalOut& e, int want_in, LdsF64* T, int sub) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    static_assert(LPE <= 2, "the fork form");
    using L = ExactLds<D, LPE>;
    CK<D> c = scalar_ref(c_in); // (a call: see scalar_ref)
    PK p = scalar_ref(p_in);
    const GoalK g = g_in; // (in registers for every pose cost of the call: through the reference it was re-read each time)
    const int want = scalar_int(want_in);
    const uint32_t pris = UZ ? 0u : c.prismatic_mask, kinds = UZ == 1 ? 0u : c.axis_kind;
    const double h = p.step_size;
    (void)sub;
    // L1P: the point is the one in rows Q0 of the lane's column (the descent keeps it there), `q_in` is not read
    constexpr bool L1P = LPE == 1 && PIK_EXACT_PAIRED && (OCC == 1 || PIK_XPAIR_OCC2) && PIK_XLDS_STATE && D <= PIK_XFORK_INLINE_MAXD;
    double qv[D];
    if constexpr (L1P) {
#pragma unroll
        for (int j = 0; j < D; ++j) qv[j] = T[(L::Q0 + j) * WAVE];
    }
    const double(&q)[D] = *reinterpret_cast<const double(*)[D]>(L1P ? static_cast<const double*>(qv) : static_cast<const double*>(q_in));
    // (unrolled: D independent polynomial chains for the scheduler to interleave)
    double qf[D];
    folded_all<D>(c.mt, q, qf); // (sincos_f64's fold for all D under one branch, pik_math.hpp)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double sn = 0.0, cs = 1.0;
        if (UZ || !((pris >> j) & 1u)) sincos_f64<false>(c.mt, qf[j], sn, cs);
        T[(L::SN0 + j) * WAVE] = sn;
        T[(L::CS0 + j) * WAVE] = cs;
    }
    double R[9], t[3];
    R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
    R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
    R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
    t[0] = t[1] = t[2] = 0.0;
    bool blank = true;
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        x_origin<D, UZ>(c, j, R, t, blank);
        const bool pj = (pris >> j) & 1u;
        const uint32_t kj = (kinds >> (2 * j)) & 3u;
        if (want) {
            // the probes of variable j branch off here: (R, t) is the frame in front of joint j
            if constexpr (LPE == 1 && PIK_EXACT_PAIRED && (OCC == 1 || PIK_XPAIR_OCC2)) {
                // one lane per elite: the - h and the + h probe walk the rest of the chain TOGETHER -- the same
                // joints, the same constants (loaded once per joint instead of twice), two independent
                // chains of arithmetic in one loop body
                double Ra[9], ta[3], Rb[9], tb[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) Ra[k] = Rb[k] = R[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) ta[k] = tb[k] = t[k];
                const double qj = L1P ? T[(L::Q0 + j) * WAVE] : q[j]; // (j is the rolled loop's counter)
                const double va = qj - h, vb = qj + h;
                double sna = 0.0, csa = 1.0, snb = 0.0, csb = 1.0;
                if (UZ || !pj) {
                    double fa = va, fb = vb;
                    folded2(c.mt, fa, fb);
                    sincos_f64<false>(c.mt, fa, sna, csa);
                    sincos_f64<false>(c.mt, fb, snb, csb);
                }
                x_joint_pair<D, UZ>(c, j, Ra, ta, Rb, tb, pj, kj, va, sna, csa, vb, snb, csb);
#pragma unroll 1
                for (int k = j + 1; k < D; ++k) {
                    const bool pk = (pris >> k) & 1u;
                    const uint32_t kk = (kinds >> (2 * k)) & 3u;
                    const double qk = L1P ? T[(L::Q0 + k) * WAVE] : q[k], snk = T[(L::SN0 + k) * WAVE], csk = T[(L::CS0 + k) * WAVE];
                    x_origin_pair<D, UZ>(c, k, Ra, ta, Rb, tb);
                    x_joint_pair<D, UZ>(c, k, Ra, ta, Rb, tb, pk, kk, qk, snk, csk, qk, snk, csk);
                }
                x_tip_pair<D, UZ>(c, Ra, ta, Rb, tb);
                EvalOut e2;
                double d2[4];
                pose_tail<D, true, NG>(c, p, g, seed, q, Ra, ta, e2, d2, j, -h);
                const double cm = e2.cost;
                if constexpr (!L1P) T[(L::CM0 + j) * WAVE] = cm;
                pose_tail<D, true, NG>(c, p, g, seed, q, Rb, tb, e2, d2, j, h);
                if constexpr (L1P) T[(L::G0 + j) * WAVE] = e2.cost - cm; // (src/ik_gradient.cpp:28-43: the difference itself)
                else T[(L::CP0 + j) * WAVE] = e2.cost;
            } else {
            constexpr int NS = LPE == 2 ? 1 : 2;
#pragma unroll 1
            for (int it = 0; it < NS; ++it) {
                const int sg = LPE == 2 ? (sub & 1) : it;
                const double dh = sg ? h : -h;
                double R2[9], t2[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) R2[k] = R[k];
                t2[0] = t[0];
                t2[1] = t[1];
                t2[2] = t[2];
                const double vj = q[j] + dh;
                double sn = 0.0, cs = 1.0;
                if (UZ || !pj) sincos_f64<false>(c.mt, folded(c.mt, vj), sn, cs);
                x_joint<D, UZ>(c, j, R2, t2, pj, kj, vj, sn, cs);
#pragma unroll 1
                for (int k = j + 1; k < D; ++k) {
                    x_origin<D, UZ>(c, k, R2, t2, false);
                    x_joint<D, UZ>(c, k, R2, t2, (pris >> k) & 1u, (kinds >> (2 * k)) & 3u, q[k],
                                   T[(L::SN0 + k) * WAVE], T[(L::CS0 + k) * WAVE]);
                }
                x_tip<D, UZ>(c, R2, t2);
                EvalOut e2;
                double d2[4];
                pose_tail<D, true, NG>(c, p, g, seed, q, R2, t2, e2, d2, j, dh);
                T[((sg ? L::CP0 : L::CM0) + j) * WAVE] = e2.cost;
            }
            }
        }
        x_joint<D, UZ>(c, j, R, t, pj, kj, (L1P && !UZ) ? T[(L::Q0 + j) * WAVE] : q[j], T[(L::SN0 + j) * WAVE], T[(L::CS0 + j) * WAVE]);
        blank = false;
    }
    x_tip<D, UZ>(c, R, t);
    double d0[4];
    pose_tail<D, false, NG>(c, p, g, seed, q, R, t, e, d0);
}
template <int D, int LPE, int OCC = 1, int UZ = 0>
__device__ __noinline__ void exact_accept_call(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], EvalOut& e,
                                          int want_in, LdsF64* T, int sub) {
    exact_accept_impl<D, LPE, OCC, UZ>(c_in, p_in, g_in, seed, q, e, want_in, T, sub);
}
template <int D, int LPE, int OCC = 1, int UZ = 0>
__device__ __forceinline__ void exact_accept(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], EvalOut& e,
                                          int want_in, LdsF64* T, int sub) {
    if constexpr (D <= PIK_XFORK_INLINE_MAXD) exact_accept_impl<D, LPE, OCC, UZ>(c_in, p_in, g_in, seed, q, e, want_in, T, sub);
    else exact_accept_call<D, LPE, OCC, UZ>(c_in, p_in, g_in, seed, q, e, want_in, T, sub);
}

// The two line-search evaluations of a step at one lane per elite (q - g and q + g, src/ik_gradient.cpp:56-64)
// walked TOGETHER, as the probe pairs above: the same joints, the constants of a joint loaded once for both, two
// independent chains of arithmetic per loop body.  Each is, operation for operation, what `evaluate` computes.
struct CostPair {
    double a, b;
};
// T (L1P, see ExactLds::Q0): the two points are q - g and q + g with q, g in rows Q0 / G0 of the lane's column, formed
// joint by joint in the rolled walk (as arrays they lived in scratch: a load and a wait per joint); qa / qb are then only
// read by the joint goals
template <int D, int OCC = 1, int UZG = 0>
__device__ __forceinline__ CostPair exact_line_pair_impl(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                                 const double (&qa)[D], const double (&qb)[D], const LdsF64* T = nullptr) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const GoalK g = g_in; // (in registers for every pose cost of the call: through the reference it was re-read each time)
    const uint32_t pris = UZ ? 0u : c.prismatic_mask, kinds = UZ == 1 ? 0u : c.axis_kind;
    double Ra[9], ta[3], Rb[9], tb[3];
    Ra[0] = 1.0; Ra[1] = 0.0; Ra[2] = 0.0;
    Ra[3] = 0.0; Ra[4] = 1.0; Ra[5] = 0.0;
    Ra[6] = 0.0; Ra[7] = 0.0; Ra[8] = 1.0;
    ta[0] = ta[1] = ta[2] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) Rb[k] = Ra[k];
    tb[0] = tb[1] = tb[2] = 0.0;
    bool blank = true;
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        const bool pj = (pris >> j) & 1u;
        const uint32_t kj = (kinds >> (2 * j)) & 3u;
        constexpr bool L1P = PIK_EXACT_PAIRED && (OCC == 1 || PIK_XPAIR_OCC2) && PIK_XLDS_STATE && D <= PIK_XFORK_INLINE_MAXD;
        double va, vb;
        if constexpr (L1P) {
            using L = ExactLds<D, 1>;
            const double qj = T[(L::Q0 + j) * WAVE], gj = T[(L::G0 + j) * WAVE];
            va = qj - gj;
            vb = qj + gj;
        } else {
            va = qa[j];
            vb = qb[j];
        }
        double sna = 0.0, csa = 1.0, snb = 0.0, csb = 1.0;
        if (UZ || !pj) {
            double fa = va, fb = vb;
            folded2(c.mt, fa, fb);
            sincos_f64<false>(c.mt, fa, sna, csa);
            sincos_f64<false>(c.mt, fb, snb, csb);
        }
        if (blank) {
            x_origin<D, UZ>(c, j, Ra, ta, true);
            x_origin<D, UZ>(c, j, Rb, tb, true);
        } else {
            x_origin_pair<D, UZ>(c, j, Ra, ta, Rb, tb);
        }
        x_joint_pair<D, UZ>(c, j, Ra, ta, Rb, tb, pj, kj, va, sna, csa, vb, snb, csb);
        blank = false;
    }
    x_tip_pair<D, UZ>(c, Ra, ta, Rb, tb);
    EvalOut e;
    double d0[4];
    CostPair out;
    pose_tail<D, false, NG>(c, p, g, seed, qa, Ra, ta, e, d0);
    out.a = e.cost;
    pose_tail<D, false, NG>(c, p, g, seed, qb, Rb, tb, e, d0);
    out.b = e.cost;
    return out;
}
template <int D, int OCC = 1, int UZ = 0>
__device__ __noinline__ CostPair exact_line_pair_call(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&qa)[D],
                                              const double (&qb)[D], const LdsF64* T) {
    return exact_line_pair_impl<D, OCC, UZ>(c_in, p_in, g_in, seed, qa, qb, T);
}
template <int D, int OCC = 1, int UZ = 0>
__device__ __forceinline__ CostPair exact_line_pair(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&qa)[D],
                                              const double (&qb)[D], const LdsF64* T) {
    if constexpr (D <= PIK_XFORK_INLINE_MAXD) return exact_line_pair_impl<D, OCC, UZ>(c_in, p_in, g_in, seed, qa, qb, T);
    else return exact_line_pair_call<D, OCC, UZ>(c_in, p_in, g_in, seed, qa, qb, T);
}

// ---- LPE >= 4: evaluations by a TEAM of lanes that hold the same joint vector ---------------------------
// The few problems that run all their generations are a chain of evaluations one after the other, and a lone
// wavefront pays every latency of that chain in full; the lanes of an elite (the accept evaluation) or the
// half of them that evaluates the same line-search point hold identical inputs, so they SHARE the part of an
// evaluation that is a long dependent chain per joint: lane r of the team takes the sine / cosine of joint r
// (+ C, + 2C ...) -- one polynomial deep instead of D -- and leaves it in the team's LDS block, where every lane
// reads all D.  Same function, same argument, same bits as computing it oneself.  The chain itself is then
// walked by every lane (the operands of joint j + 1 requested while joint j is computed).

// the chain constants of one joint (scalar registers)
struct JointConsts {
    double o[12]; // origin transform
};
template <int D>
__device__ __forceinline__ void load_joint_consts(CK<D> c, int j, JointConsts& k) {
    CPtr o = c.O[j];
#pragma unroll
    for (int i = 0; i < 12; ++i) k.o[i] = o[i];
}
// (R, t) <- (R, t) * origin held in registers; skipped when it is exactly the identity, copied while blank
template <int D>
__device__ __forceinline__ void chain_origin_r(CK<D> c, int j, const JointConsts& k, double (&R)[9], double (&t)[3],
                                               bool blank) {
    if ((c.origin_ident_mask >> j) & 1u) return;
    if (blank) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = k.o[i];
        t[0] = k.o[9];
        t[1] = k.o[10];
        t[2] = k.o[11];
    } else {
        iso_mul_r(R, t, k.o);
    }
}

// ---- one ROW of the running frame through the literal chain product ------------------------------------
// (R, t) * M needs, for row i of the result, row i of R and t[i] only, and every element is the same sequence of
// operations on the same operands as in iso_mul / rotate_exact / chain_joint (pik_math.hpp) -- so three lanes
// that carry one row each produce, element for element, the bits one lane carrying the whole frame produces.
template <typename O>
__device__ __forceinline__ void row_iso(double (&r)[3], double& t, const O& o) {
    const double r0 = r[0], r1 = r[1], r2 = r[2];
    r[0] = xdot3(r0, o[0], r1, o[3], r2, o[6]);
    r[1] = xdot3(r0, o[1], r1, o[4], r2, o[7]);
    r[2] = xdot3(r0, o[2], r1, o[5], r2, o[8]);
#if PIK_XF
    t = fma_f64(r2, o[11], fma_f64(r1, o[10], fma_f64(r0, o[9], t)));
#else
    t = r0 * o[9] + r1 * o[10] + r2 * o[11] + t;
#endif
}
// ... by a fixed transform of a chain of class 1 / 2 (x_iso_mul, pik_math.hpp: the exact 1 and 0 entries left out)
template <int XM, typename O>
__device__ __forceinline__ void row_iso_k(double (&r)[3], double& t, const O& o, uint32_t kind, uint32_t pm) {
#if PIK_XF
    if (pm & 1u) t = fma_f64(r[0], o[9], t);
    if (pm & 2u) t = fma_f64(r[1], o[10], t);
    if (pm & 4u) t = fma_f64(r[2], o[11], t);
    if constexpr (XM == 1) {
        (void)kind;
        iso_rot_row<ISO_RX>(r[0], r[1], r[2], o);
    } else if constexpr (XM == 0 || !PIK_XSPARSE_K2) {
        (void)kind;
        iso_rot_row<ISO_GENERAL>(r[0], r[1], r[2], o);
    } else {
        if (kind == ISO_TRANS) {
        } else if (kind == ISO_RX) {
            iso_rot_row<ISO_RX>(r[0], r[1], r[2], o);
        } else if (kind == ISO_RY) {
            iso_rot_row<ISO_RY>(r[0], r[1], r[2], o);
        } else if (kind == ISO_RZ) {
            iso_rot_row<ISO_RZ>(r[0], r[1], r[2], o);
        } else {
            iso_rot_row<ISO_GENERAL>(r[0], r[1], r[2], o);
        }
    }
#else
    (void)kind;
    (void)pm;
    row_iso(r, t, o);
#endif
}
// the row times the revolute joint's rotation (rotate_exact / rotate_about's general branch, one row)
__device__ __forceinline__ void row_rotate(double (&r)[3], uint32_t kind, CPtr a, double sn, double cs) {
    const double r0 = r[0], r1 = r[1], r2 = r[2];
    if (kind == AXIS_GENERAL) {
        const double x = a[0], y = a[1], z = a[2];
        const double tt = 1.0 - cs;
        const double txy = tt * (x * y), txz = tt * (x * z), tyz = tt * (y * z);
        const double zs = z * sn, ys = y * sn, xs = x * sn;
        double J[9];
        J[0] = tt * (x * x) + cs;
        J[3] = txy + zs;
        J[6] = txz - ys;
        J[1] = txy - zs;
        J[4] = tt * (y * y) + cs;
        J[7] = tyz + xs;
        J[2] = txz + ys;
        J[5] = tyz - xs;
        J[8] = tt * (z * z) + cs;
        r[0] = xdot3(r0, J[0], r1, J[3], r2, J[6]);
        r[1] = xdot3(r0, J[1], r1, J[4], r2, J[7]);
        r[2] = xdot3(r0, J[2], r1, J[5], r2, J[8]);
        return;
    }
    const double tt = 1.0 - cs;
    const double d = tt + cs;
    if (kind == AXIS_Z) {
        r[0] = xmad(r1, sn, r0 * cs);
        r[1] = xmad(r1, cs, -(r0 * sn));
        r[2] = r2 * d;
    } else if (kind == AXIS_Y) {
        r[0] = xmad(r2, -sn, r0 * cs);
        r[1] = r1 * d;
        r[2] = xmad(r2, cs, r0 * sn);
    } else {
        r[0] = r0 * d;
        r[1] = xmad(r2, sn, r1 * cs);
        r[2] = xmad(r2, cs, r1 * (-sn));
    }
}
// ... for an axis that is exactly +x / +y / +z (UA; XM = 1: +z, UZ): row_rotate's three special cases
template <int XM>
__device__ __forceinline__ void row_rotate_axis(double (&r)[3], uint32_t kind, double sn, double cs) {
    const double r0 = r[0], r1 = r[1], r2 = r[2];
    const double tt = 1.0 - cs;
    const double d = tt + cs;
    if (XM == 1 || kind == AXIS_Z) {
        r[0] = xmad(r1, sn, r0 * cs);
        r[1] = xmad(r1, cs, -(r0 * sn));
        r[2] = r2 * d;
    } else if (kind == AXIS_Y) {
        r[0] = xmad(r2, -sn, r0 * cs);
        r[1] = r1 * d;
        r[2] = xmad(r2, cs, r0 * sn);
    } else {
        r[0] = r0 * d;
        r[1] = xmad(r2, sn, r1 * cs);
        r[2] = xmad(r2, cs, r1 * (-sn));
    }
}
template <int D>
__device__ __forceinline__ void row_joint(CK<D> c, int j, double (&r)[3], double& t, bool prismatic, uint32_t kind,
                                          double v, double sn, double cs) {
    CPtr a = c.axis[j];
    if (prismatic) {
#if PIK_XF
        t = fma_f64(r[2], a[2] * v, fma_f64(r[1], a[1] * v, fma_f64(r[0], a[0] * v, t)));
#else
        t = r[0] * (a[0] * v) + r[1] * (a[1] * v) + r[2] * (a[2] * v) + t;
#endif
    } else {
        row_rotate(r, kind, a, sn, cs);
    }
}

// One evaluation by a team of C lanes (rank r) that all hold q: cost + verdict as `evaluate`.  TB = the team's
// LDS block [sn D][cs D][q D]; STORE: the frames in front of the joints go to PF ([D][12]) when `store`, for the
// probe passes.  With three or more lanes the team also splits the chain product: lanes 0, 1, 2 carry one ROW of
// the running frame each (the others repeat one of them) -- a third of the arithmetic on the critical path -- and
// the rows of the final frame are exchanged through XF (12 numbers) for the pose cost, which every lane takes.
// TAIL = false (STORE only): no pose cost here -- the frame behind the last joint, in front of the tip transform,
// is left in XF as "frame D", and a spare lane of the probe passes finishes the evaluation beside the probes
// (exact_probe_pass).
template <int D, int C, bool STORE, bool TAIL = true, int UZG = 0>
__device__ __forceinline__ CostSol exact_eval_team_impl(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                                const double (&q)[D], LdsF64* TB, LdsF64* PF, LdsF64* XF, int r,
                                                int store_in) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    CostSol out;
    out.cost = 0.0;
    out.sol = 0;
    static_assert(TAIL || (STORE && C >= 3), "the evaluation without its pose cost: the accept evaluation of a wide elite");
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const GoalK g = g_in; // (in registers for every pose cost of the call: through the reference it was re-read each time)
    const uint32_t pris = c.prismatic_mask, kinds = c.axis_kind;
    const bool store = store_in != 0;
    (void)PF;
    (void)XF;
    (void)store;
    constexpr int KP = (D + C - 1) / C;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int j = k * C + r;
        const int jj = j < D ? j : D - 1;
        double qv = q[0];
#pragma unroll
        for (int i = 1; i < D; ++i) qv = (jj == i) ? q[i] : qv;
        double sn, cs;
        sincos_f64<false>(c.mt, folded(c.mt, qv), sn, cs); // (a prismatic joint does not use it)
        if (j < D) {
            TB[jj] = sn;
            TB[D + jj] = cs;
            TB[2 * D + jj] = qv;
        }
    }
    wave_sync();
    double R[9], t[3];
    if constexpr (UZ) {
        // every joint a rotation about z behind a non-identity origin: the loop over the joints unrolled
        if constexpr (C >= 3) {
            const int row = r - 3 * (r / 3);
            const bool writer = r < 3;
            double rr[3], tr;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                CPtr o = c.O[j];
                const double sn = TB[j], cs = TB[D + j];
                if (j == 0) { // nothing multiplied in yet: the origin is copied
#pragma unroll
                    for (int k = 0; k < 3; ++k) rr[k] = row == 0 ? o[k] : row == 1 ? o[3 + k] : o[6 + k];
                    tr = row == 0 ? o[9] : row == 1 ? o[10] : o[11];
                } else {
                    row_iso_k<UZ>(rr, tr, o, PIK_OKIND(c, j), PIK_OPM(c, j));
                }
                if constexpr (STORE) {
                    if (store && writer) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) PF[12 * j + 3 * row + k] = rr[k];
                        PF[12 * j + 9 + row] = tr;
                    }
                }
                row_rotate_axis<UZ>(rr, (kinds >> (2 * j)) & 3u, sn, cs);
            }
            if constexpr (TAIL) {
                if constexpr (UZ == 1 && PIK_XF) { // (a z twist: x_tip_mul<1>, one row)
                    tr = fma_f64(rr[2], c.tip[11], fma_f64(rr[1], c.tip[10], fma_f64(rr[0], c.tip[9], tr)));
                    iso_rot_row<ISO_RZ>(rr[0], rr[1], rr[2], c.tip);
                } else {
                    row_iso_k<UZ == 2 ? 2 : 0>(rr, tr, c.tip, c.tip_kind, PIK_TPM(c));
                }
            }
            if (writer && (TAIL || store)) {
#pragma unroll
                for (int k = 0; k < 3; ++k) XF[3 * row + k] = rr[k];
                XF[9 + row] = tr;
            }
            if constexpr (!TAIL) return out;
            wave_sync();
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = XF[k];
            t[0] = XF[9];
            t[1] = XF[10];
            t[2] = XF[11];
        } else {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                x_origin<D, UZ>(c, j, R, t, j == 0);
                if constexpr (STORE) {
                    if (store && r == 0) {
#pragma unroll
                        for (int k = 0; k < 9; ++k) PF[12 * j + k] = R[k];
                        PF[12 * j + 9] = t[0];
                        PF[12 * j + 10] = t[1];
                        PF[12 * j + 11] = t[2];
                    }
                }
                x_rotate<UZ>(R, (kinds >> (2 * j)) & 3u, TB[j], TB[D + j]);
            }
            x_tip<D, UZ>(c, R, t);
        }
        double d0u[4];
        EvalOut eu;
        pose_tail<D, false, NG>(c, p, g, seed, q, R, t, eu, d0u);
        out.cost = eu.cost;
        out.sol = eu.sol ? 1 : 0;
        return out;
    }
    JointConsts kn;
    load_joint_consts<D>(c, 0, kn);
    double sn_n = TB[0], cs_n = TB[D], v_n = TB[2 * D];
    if constexpr (C >= 3) {
        const int row = r - 3 * (r / 3);
        const bool writer = r < 3;
        double rr[3] = {row == 0 ? 1.0 : 0.0, row == 1 ? 1.0 : 0.0, row == 2 ? 1.0 : 0.0};
        double tr = 0.0;
#pragma unroll 1
        for (int j = 0; j < D; ++j) {
            const JointConsts kc = kn;
            const double sn = sn_n, cs = cs_n, v = v_n;
            const int jn = j + 1 < D ? j + 1 : j; // the next joint's operands: in flight during this joint
            load_joint_consts<D>(c, jn, kn);
            sn_n = TB[jn];
            cs_n = TB[D + jn];
            v_n = TB[2 * D + jn];
            if (!((c.origin_ident_mask >> j) & 1u)) {
                if (j == 0) { // nothing multiplied in yet: the origin is copied (chain_origin_r, blank)
#pragma unroll
                    for (int k = 0; k < 3; ++k) rr[k] = row == 0 ? kc.o[k] : row == 1 ? kc.o[3 + k] : kc.o[6 + k];
                    tr = row == 0 ? kc.o[9] : row == 1 ? kc.o[10] : kc.o[11];
                } else {
                    row_iso(rr, tr, kc.o);
                }
            }
            if constexpr (STORE) {
                if (store && writer) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) PF[12 * j + 3 * row + k] = rr[k];
                    PF[12 * j + 9 + row] = tr;
                }
            }
            row_joint<D>(c, j, rr, tr, (pris >> j) & 1u, (kinds >> (2 * j)) & 3u, v, sn, cs);
        }
        if constexpr (TAIL) {
            if (!c.tip_ident) row_iso(rr, tr, c.tip);
        }
        if (writer && (TAIL || store)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) XF[3 * row + k] = rr[k];
            XF[9 + row] = tr;
        }
        if constexpr (!TAIL) return out;
        wave_sync();
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = XF[k];
        t[0] = XF[9];
        t[1] = XF[10];
        t[2] = XF[11];
    } else {
        R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
        R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
        R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
        t[0] = t[1] = t[2] = 0.0;
        bool blank = true;
#pragma unroll 1
        for (int j = 0; j < D; ++j) {
            const JointConsts kc = kn;
            const double sn = sn_n, cs = cs_n, v = v_n;
            const int jn = j + 1 < D ? j + 1 : j;
            load_joint_consts<D>(c, jn, kn);
            sn_n = TB[jn];
            cs_n = TB[D + jn];
            v_n = TB[2 * D + jn];
            chain_origin_r<D>(c, j, kc, R, t, blank);
            if constexpr (STORE) {
                if (store && r == 0) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) PF[12 * j + k] = R[k];
                    PF[12 * j + 9] = t[0];
                    PF[12 * j + 10] = t[1];
                    PF[12 * j + 11] = t[2];
                }
            }
            chain_joint<D>(c, j, R, t, (pris >> j) & 1u, (kinds >> (2 * j)) & 3u, v, sn, cs);
            blank = false;
        }
        if (!c.tip_ident) iso_mul(R, t, c.tip);
    }
    double d0[4];
    EvalOut e;
    pose_tail<D, false, NG>(c, p, g, seed, q, R, t, e, d0);
    out.cost = e.cost;
    out.sol = e.sol ? 1 : 0;
    return out;
}
template <int D, int C, bool STORE, bool TAIL = true, int UZ = 0>
__device__ __noinline__ CostSol exact_eval_team_call(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], LdsF64* TB, LdsF64* PF,
                                                LdsF64* XF, int r, int store_in) {
    return exact_eval_team_impl<D, C, STORE, TAIL, UZ>(c_in, p_in, g_in, seed, q, TB, PF, XF, r, store_in);
}
template <int D, int C, bool STORE, bool TAIL = true, int UZ = 0>
__device__ __forceinline__ CostSol exact_eval_team(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], LdsF64* TB, LdsF64* PF,
                                                LdsF64* XF, int r, int store_in) {
    if constexpr (D <= PIK_XTEAM_INLINE_MAXD) return exact_eval_team_impl<D, C, STORE, TAIL, UZ>(c_in, p_in, g_in, seed, q, TB, PF, XF, r, store_in);
    else return exact_eval_team_call<D, C, STORE, TAIL, UZ>(c_in, p_in, g_in, seed, q, TB, PF, XF, r, store_in);
}

// The address of a joint's constants BEHIND a vector value: the scalar loads through the returned pointer cannot be
// issued before `dep` exists.  The flat walks request joint j + 1's constants when joint j begins (they land while it
// computes; two joints' worth of scalar registers in flight) -- left to itself the scheduler hoists every joint's
// loads to the top of the unrolled walk, a hundred scalar registers that spill into vector lanes (two v_readlane per
// operand at every use).
template <typename T>
__device__ __forceinline__ const PIK_CONSTANT T* x_after(const PIK_CONSTANT T* p, double dep) {
    asm volatile("" : "+s"(p) : "v"(dep));
    return p;
}

// The sines / cosines of q by a team of C lanes, left in the team's block TB = [sn D][cs D][q D] (the first phase of
// exact_eval_team, by itself: a flat probe pass that holds every probe AND the accept evaluation needs nothing else of it)
template <int D, int C>
__device__ __forceinline__ void exact_team_sincos(CK<D> c_in, const double (&q)[D], LdsF64* TB, int r) {
    CK<D> c = scalar_ref(c_in);
    constexpr int KP = (D + C - 1) / C;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int j = k * C + r;
        const int jj = j < D ? j : D - 1;
        double qv = q[0];
#pragma unroll
        for (int i = 1; i < D; ++i) qv = (jj == i) ? q[i] : qv;
        double sn, cs;
        sincos_f64<false>(c.mt, folded(c.mt, qv), sn, cs);
        if (j < D) {
            TB[jj] = sn;
            TB[D + jj] = cs;
            TB[2 * D + jj] = qv;
        }
    }
    wave_sync();
}

// One pass of probes at LPE >= 4 lanes per elite: lane `sub` evaluates probe `probe + sub` (2 i -> q - h e_i,
// 2 i + 1 -> q + h e_i; a lane beyond 2D: the last joint with no displacement, result unused) from the
// frame in front of ITS joint i (PF) with the other joints' sines / cosines of the accept evaluation (EB); the
// joints from the pass's first joint to the tip are walked in lock-step (a lane waits until the walk reaches its
// joint).  Returns the probe's cost.  `fused`: the lane of "probe" 2D finishes the ACCEPT evaluation -- it starts
// from frame D (behind the last joint), walks nothing, multiplies the tip transform in and takes the pose cost of
// q itself, in the instructions the probes spend on theirs anyway; it returns that cost and verdict.
// FLAT (chains of class 1 / 2; see PIK_XFLAT): 1 = every lane starts from the frame in front of the pass's FIRST joint,
// 2 = from nothing (the pass holds every probe: its first joint is joint 0, whose origin is copied), and walks every
// joint of the pass; the lane of the accept evaluation (`fused`) walks them with the accept evaluation's values.
template <int D, int LPE, int UZG = 0, int FLAT = 0>
__device__ __forceinline__ CostSol exact_probe_pass_impl(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                                 const double (&q)[D], int probe_in, const LdsF64* EB,
                                                 const LdsF64* PF, int sub, int fused_in) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const GoalK g = g_in; // (in registers for every pose cost of the call: through the reference it was re-read each time)
    const int probe = scalar_int(probe_in);
    const int fused = scalar_int(fused_in);
    const uint32_t pris = c.prismatic_mask, kinds = c.axis_kind;
    const double h = p.step_size;
    const int pr = probe + sub;
    const bool valid = pr < 2 * D;
    const int i = valid ? (pr >> 1) : ((fused && pr == 2 * D) ? D : D - 1);
    const double dh = valid ? ((pr & 1) ? h : -h) : 0.0;
    const int jmin = probe >> 1; // wave-uniform; every lane's joint is >= jmin
    constexpr int FL = UZ ? FLAT : 0;
    const int f0 = FL ? jmin : i; // the frame a lane starts from
    double R[9], t[3];
    if constexpr (FL != 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = PF[12 * f0 + k];
        t[0] = PF[12 * f0 + 9];
        t[1] = PF[12 * f0 + 10];
        t[2] = PF[12 * f0 + 11];
    }
    double vi = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) vi = (k == i) ? q[k] + dh : vi;
    double sni = 0.0, csi = 1.0;
    sincos_f64<false>(c.mt, folded(c.mt, vi), sni, csi); // (unused by a prismatic joint)
    if constexpr (UZ && FL == 2) {
        // the whole chain, no lane predicated: the accepted point's sines / cosines into registers first, the joints'
        // constants one joint ahead (x_after)
        // (class 1 only -- PRE: a walk of class 2 keeps its per-joint decisions, and with everything requested ahead
        //  of them the descent's register count doubles)
        constexpr bool PRE = UZ == 1;
        double esn[D], ecs[D];
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                esn[j] = EB[j];
                ecs[j] = EB[D + j];
            }
        }
        CPtr on = c.O[D > 1 ? 1 : 0];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            CPtr o = (j == 0 || !PRE) ? c.O[j] : on;
            if (PRE && j >= 1 && j + 1 < D) on = x_after(c.O[j + 1], R[0]);
            if (j == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = o[k];
                t[0] = o[9];
                t[1] = o[10];
                t[2] = o[11];
            } else {
                x_iso_mul<UZ>(R, t, o, PIK_OKIND(c, j), PIK_OPM(c, j));
            }
            const bool own = j == i;
            x_rotate<UZ>(R, (kinds >> (2 * j)) & 3u, own ? sni : (PRE ? esn[j] : EB[j]), own ? csi : (PRE ? ecs[j] : EB[D + j]));
        }
        x_tip<D, UZ>(c, R, t);
        EvalOut eu;
        double du[4];
        pose_tail<D, true, NG>(c, p, g, seed, q, R, t, eu, du, i, dh);
        CostSol ou;
        ou.cost = eu.cost;
        ou.sol = eu.sol ? 1 : 0;
        return ou;
    } else if constexpr (UZ) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j < jmin) continue; // (wave-uniform)
            if (FL ? (j > jmin) : (j > i)) x_iso_mul<UZ>(R, t, c.O[j], PIK_OKIND(c, j), PIK_OPM(c, j));
            if (FL || j >= i) {
                const bool own = j == i;
                x_rotate<UZ>(R, (kinds >> (2 * j)) & 3u, own ? sni : EB[j], own ? csi : EB[D + j]);
            }
        }
        x_tip<D, UZ>(c, R, t);
        EvalOut eu;
        double du[4];
        pose_tail<D, true, NG>(c, p, g, seed, q, R, t, eu, du, i, dh);
        CostSol ou;
        ou.cost = eu.cost;
        ou.sol = eu.sol ? 1 : 0;
        return ou;
    }
    JointConsts kn;
    load_joint_consts<D>(c, jmin, kn);
    double sn_n = EB[jmin], cs_n = EB[D + jmin], v_n = EB[2 * D + jmin];
#pragma unroll 1
    for (int j = jmin; j < D; ++j) {
        const JointConsts kc = kn;
        const double sn_c = sn_n, cs_c = cs_n, v_c = v_n;
        const int jn = j + 1 < D ? j + 1 : j;
        load_joint_consts<D>(c, jn, kn);
        sn_n = EB[jn];
        cs_n = EB[D + jn];
        v_n = EB[2 * D + jn];
        if (j > i) chain_origin_r<D>(c, j, kc, R, t, false);
        if (j >= i) {
            const bool own = j == i;
            chain_joint<D>(c, j, R, t, (pris >> j) & 1u, (kinds >> (2 * j)) & 3u, own ? vi : v_c, own ? sni : sn_c,
                           own ? csi : cs_c);
        }
    }
    if (!c.tip_ident) iso_mul(R, t, c.tip);
    EvalOut e2;
    double d2[4];
    pose_tail<D, true, NG>(c, p, g, seed, q, R, t, e2, d2, i, dh);
    CostSol out;
    out.cost = e2.cost;
    out.sol = e2.sol ? 1 : 0;
    return out;
}
template <int D, int LPE, int UZ = 0>
__device__ __noinline__ CostSol exact_probe_pass_call(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], int probe_in,
                                                 const LdsF64* EB, const LdsF64* PF, int sub, int fused_in) {
    return exact_probe_pass_impl<D, LPE, UZ>(c_in, p_in, g_in, seed, q, probe_in, EB, PF, sub, fused_in);
}
template <int D, int LPE, int UZ = 0, int FLAT = 0>
__device__ __forceinline__ CostSol exact_probe_pass(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], int probe_in,
                                                 const LdsF64* EB, const LdsF64* PF, int sub, int fused_in) {
    if constexpr (D <= PIK_XTEAM_INLINE_MAXD) return exact_probe_pass_impl<D, LPE, UZ, FLAT>(c_in, p_in, g_in, seed, q, probe_in, EB, PF, sub, fused_in);
    else return exact_probe_pass_call<D, LPE, UZ>(c_in, p_in, g_in, seed, q, probe_in, EB, PF, sub, fused_in);
}

// The same pass with TWO probes per lane: lane `sub` takes joint i = joint0 + sub and evaluates q - h e_i AND
// q + h e_i side by side -- the same start frame, the same joints behind it, each joint's constants and the
// accept evaluation's sines fetched once for both (as the one-lane fork does, exact_accept).  Half the passes at
// ~1.45x the cost of one.  `fused`: the lane of "joint" D finishes the accept evaluation (first member of its
// pair; see exact_probe_pass).
struct CostPairSol {
    double a, b; // cost of q - h e_i, of q + h e_i  (accept lane: a = the cost of q)
    int sol;     // verdict of the first member
};
// J0C >= 0: the pass's first joint as a compile-time constant (the flat passes of a descent whose pass loop is
// unrolled: no decision about a joint is left at run time)
template <int D, int LPE, int UZG = 0, int FLAT = 0, int J0C = -1>
__device__ __forceinline__ CostPairSol exact_probe_pair_impl(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                                     const double (&q)[D], int joint0_in, const LdsF64* EB,
                                                     const LdsF64* PF, int sub, int fused_in) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const GoalK g = g_in; // (in registers for every pose cost of the call: through the reference it was re-read each time)
    const int joint0 = J0C >= 0 ? J0C : scalar_int(joint0_in);
    const int fused = scalar_int(fused_in);
    const uint32_t pris = c.prismatic_mask, kinds = c.axis_kind;
    const double h = p.step_size;
    const int slot = joint0 + sub;
    const bool valid = slot < D;
    const int i = valid ? slot : ((fused && slot == D) ? D : D - 1);
    const double hm = valid ? -h : 0.0, hp = valid ? h : 0.0;
    // (see exact_probe_pass_impl; a flat pass whose first joint is joint 0 starts from nothing: the origin is copied)
    constexpr int FL = UZ ? ((FLAT == 1 && J0C == 0) ? 2 : FLAT) : 0;
    static_assert(FL != 2 || J0C <= 0, "a pass that starts from nothing starts at joint 0");
    const int f0 = FL ? joint0 : i;
    double Ra[9], ta[3], Rb[9], tb[3];
    if constexpr (FL != 2) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Ra[k] = Rb[k] = PF[12 * f0 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) ta[k] = tb[k] = PF[12 * f0 + 9 + k];
    }
    double qi = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) qi = (k == i) ? q[k] : qi;
    const double va = qi + hm, vb = qi + hp;
    double sna = 0.0, csa = 1.0, snb = 0.0, csb = 1.0;
    {
        double fa = va, fb = vb;
        folded2(c.mt, fa, fb);
        sincos_f64<false>(c.mt, fa, sna, csa); // (unused by a prismatic joint)
        sincos_f64<false>(c.mt, fb, snb, csb);
    }
    if constexpr (UZ && FL == 2) {
        // (see exact_probe_pass_impl)
        constexpr bool PRE = UZ == 1;
        double esn[D], ecs[D];
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                esn[j] = EB[j];
                ecs[j] = EB[D + j];
            }
        }
        CPtr on = c.O[D > 1 ? 1 : 0];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            CPtr o = (j == 0 || !PRE) ? c.O[j] : on;
            if (PRE && j >= 1 && j + 1 < D) on = x_after(c.O[j + 1], Ra[0]);
            if (j == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) Ra[k] = Rb[k] = o[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) ta[k] = tb[k] = o[9 + k];
            } else {
                x_iso_mul_pair<UZ>(Ra, ta, Rb, tb, o, PIK_OKIND(c, j), PIK_OPM(c, j));
            }
            const bool own = j == i;
            const double sn_c = PRE ? esn[j] : EB[j], cs_c = PRE ? ecs[j] : EB[D + j];
            x_rotate_pair<UZ>(Ra, Rb, (kinds >> (2 * j)) & 3u, own ? sna : sn_c, own ? csa : cs_c, own ? snb : sn_c,
                              own ? csb : cs_c);
        }
        x_tip_pair<D, UZ>(c, Ra, ta, Rb, tb);
        EvalOut eu;
        double du[4];
        CostPairSol ou;
        pose_tail<D, true, NG>(c, p, g, seed, q, Ra, ta, eu, du, i, hm);
        ou.a = eu.cost;
        ou.sol = eu.sol ? 1 : 0;
        pose_tail<D, true, NG>(c, p, g, seed, q, Rb, tb, eu, du, i, hp);
        ou.b = eu.cost;
        return ou;
    } else if constexpr (UZ && FL == 1 && J0C > 0) {
        // a later flat pass with its first joint known: joints J0C .. D - 1 from the stored frame in front of joint J0C
        static_assert(J0C < D || J0C < 0, "a pass of probes has a joint");
        constexpr int J0 = J0C > 0 ? J0C : 0;
        constexpr bool PRE = UZ == 1;
        double esn[D], ecs[D];
        if constexpr (PRE) {
#pragma unroll
            for (int j = J0; j < D; ++j) {
                esn[j] = EB[j];
                ecs[j] = EB[D + j];
            }
        }
        CPtr on = c.O[J0 + 1 < D ? J0 + 1 : J0];
#pragma unroll
        for (int j = J0; j < D; ++j) {
            CPtr o = PRE ? on : c.O[j];
            if (PRE && j > J0 && j + 1 < D) on = x_after(c.O[j + 1], Ra[0]);
            if (j > J0) x_iso_mul_pair<UZ>(Ra, ta, Rb, tb, o, PIK_OKIND(c, j), PIK_OPM(c, j));
            const bool own = j == i;
            const double sn_c = PRE ? esn[j] : EB[j], cs_c = PRE ? ecs[j] : EB[D + j];
            x_rotate_pair<UZ>(Ra, Rb, (kinds >> (2 * j)) & 3u, own ? sna : sn_c, own ? csa : cs_c, own ? snb : sn_c,
                              own ? csb : cs_c);
        }
        x_tip_pair<D, UZ>(c, Ra, ta, Rb, tb);
        EvalOut eu;
        double du[4];
        CostPairSol ou;
        pose_tail<D, true, NG>(c, p, g, seed, q, Ra, ta, eu, du, i, hm);
        ou.a = eu.cost;
        ou.sol = eu.sol ? 1 : 0;
        pose_tail<D, true, NG>(c, p, g, seed, q, Rb, tb, eu, du, i, hp);
        ou.b = eu.cost;
        return ou;
    } else if constexpr (UZ) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j < joint0) continue; // (wave-uniform)
            if (FL ? (j > joint0) : (j > i)) {
                x_iso_mul_pair<UZ>(Ra, ta, Rb, tb, c.O[j], PIK_OKIND(c, j), PIK_OPM(c, j));
            }
            if (FL || j >= i) {
                const bool own = j == i;
                const double sn_c = EB[j], cs_c = EB[D + j];
                x_rotate_pair<UZ>(Ra, Rb, (kinds >> (2 * j)) & 3u, own ? sna : sn_c, own ? csa : cs_c, own ? snb : sn_c,
                                  own ? csb : cs_c);
            }
        }
        x_tip_pair<D, UZ>(c, Ra, ta, Rb, tb);
        EvalOut eu;
        double du[4];
        CostPairSol ou;
        pose_tail<D, true, NG>(c, p, g, seed, q, Ra, ta, eu, du, i, hm);
        ou.a = eu.cost;
        ou.sol = eu.sol ? 1 : 0;
        pose_tail<D, true, NG>(c, p, g, seed, q, Rb, tb, eu, du, i, hp);
        ou.b = eu.cost;
        return ou;
    }
    JointConsts kn;
    load_joint_consts<D>(c, joint0, kn);
    double sn_n = EB[joint0], cs_n = EB[D + joint0], v_n = EB[2 * D + joint0];
#pragma unroll 1
    for (int j = joint0; j < D; ++j) {
        const JointConsts kc = kn;
        const double sn_c = sn_n, cs_c = cs_n, v_c = v_n;
        const int jn = j + 1 < D ? j + 1 : j;
        load_joint_consts<D>(c, jn, kn);
        sn_n = EB[jn];
        cs_n = EB[D + jn];
        v_n = EB[2 * D + jn];
        if (j > i) {
            chain_origin_r<D>(c, j, kc, Ra, ta, false);
            chain_origin_r<D>(c, j, kc, Rb, tb, false);
        }
        if (j >= i) {
            const bool own = j == i;
            const bool pj = (pris >> j) & 1u;
            const uint32_t kj = (kinds >> (2 * j)) & 3u;
            chain_joint<D>(c, j, Ra, ta, pj, kj, own ? va : v_c, own ? sna : sn_c, own ? csa : cs_c);
            chain_joint<D>(c, j, Rb, tb, pj, kj, own ? vb : v_c, own ? snb : sn_c, own ? csb : cs_c);
        }
    }
    if (!c.tip_ident) {
        iso_mul(Ra, ta, c.tip);
        iso_mul(Rb, tb, c.tip);
    }
    EvalOut e2;
    double d2[4];
    CostPairSol out;
    pose_tail<D, true, NG>(c, p, g, seed, q, Ra, ta, e2, d2, i, hm);
    out.a = e2.cost;
    out.sol = e2.sol ? 1 : 0;
    pose_tail<D, true, NG>(c, p, g, seed, q, Rb, tb, e2, d2, i, hp);
    out.b = e2.cost;
    return out;
}
template <int D, int LPE, int UZ = 0>
__device__ __noinline__ CostPairSol exact_probe_pair_call(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], int joint0_in,
                                                     const LdsF64* EB, const LdsF64* PF, int sub, int fused_in) {
    return exact_probe_pair_impl<D, LPE, UZ>(c_in, p_in, g_in, seed, q, joint0_in, EB, PF, sub, fused_in);
}
template <int D, int LPE, int UZ = 0, int FLAT = 0, int J0C = -1>
__device__ __forceinline__ CostPairSol exact_probe_pair(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D], const double (&q)[D], int joint0_in,
                                                     const LdsF64* EB, const LdsF64* PF, int sub, int fused_in) {
    if constexpr (D <= PIK_XTEAM_INLINE_MAXD) return exact_probe_pair_impl<D, LPE, UZ, FLAT, J0C>(c_in, p_in, g_in, seed, q, joint0_in, EB, PF, sub, fused_in);
    else return exact_probe_pair_call<D, LPE, UZ>(c_in, p_in, g_in, seed, q, joint0_in, EB, PF, sub, fused_in);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose counter is a compile-time constant
// in every iteration's body (the flat probe passes: the first joint of each pass)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void x_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        x_static_for<N, I + 1>(f);
    }
}

// GradientIk::from + step() + the driver loops of MemeticIk::gradientDescent (GD_ELITE,
// src/ik_memetic.cpp:66-91), ik_gradient (GD_LOCAL, src/ik_gradient.cpp:96-139) and one step (GD_SINGLE):
// the same bookkeeping, statement for statement, as gradient_descent's PIK_STRICT paths.
// (a real call: the descent's registers are allocated on their own, not on top of everything the memetic
//  kernel keeps alive around it -- inlined, the kernels for 8 and more variables sat at 512 registers + scratch
//  and faulted)
template <int D, int MODE, int LPE, int OCC = 1, int UZG = 0>
__device__ __noinline__ void gradient_descent_exact(CK<D> c_in, PK p_in, const GoalK& g, const double (&seed)[D],
                                                    GdState<D>& s_io, bool active, int max_iters_in, double* lds,
                                                    int lane, int sub) {
    // UZG: the chain class (pik_math.hpp), or 3 = class 1 in a call without joint goals (PIK_GM(p) a compile-time 0)
    constexpr int UZ = UZG == 3 ? 1 : UZG;
    constexpr bool NG = UZG == 3;
    (void)NG;
    // The descent's state in registers for the length of the descent, member by member (through the reference every
    // read and write of it was a scratch access, ~80 per step with their waits on a lone wavefront's critical path;
    // as ONE local struct it stays in memory too, because the joint vector is handed to the evaluations by
    // reference and drags the whole struct with it -- 141 scratch accesses in the loop instead of 91).  The joint
    // vector has a second copy in memory, `qmem`, for the evaluations to read.
    struct {
        double local_cost, best_cost;
        bool best_sol;
        int steps, iters, found;
    } s;
    // (PIK_XGD_REGS_OCC2 = 0: the kernels compiled for two wavefronts per SIMD have no register to spare -- 256, no
    //  AGPRs -- and keep the arrays where they were, in the caller's frame)
    // ... and so do the long chains, whose descent sits at the register cap as it is (sixteen variables: 256 + 256
    // registers and the first vector spills with the arrays in registers)
    constexpr bool REGS = D <= PIK_XTEAM_INLINE_MAXD && (OCC == 1 || PIK_XGD_REGS_OCC2);
    double loc_r[D], bst_r[D], grd_r[D], qmem_r[D];
    double(&loc)[D] = REGS ? loc_r : s_io.local;
    double(&bst)[D] = REGS ? bst_r : s_io.best;
    double(&grd)[D] = REGS ? grd_r : s_io.grad;
    double(&qmem)[D] = REGS ? qmem_r : s_io.local;
    // L1P (one lane per elite, probes in pairs): the point lives in rows Q0 of the lane's LDS column, the gradient in
    // rows G0 (ExactLds) -- `loc` / `qmem` are not used, `grd` / `bst` are only written
    constexpr bool L1P = LPE == 1 && PIK_EXACT_PAIRED && (OCC == 1 || PIK_XPAIR_OCC2) && PIK_XLDS_STATE && D <= PIK_XFORK_INLINE_MAXD;
    if constexpr (L1P) {
        double l0[D]; // (all the loads first: through a generic pointer they may not pass an LDS store)
#pragma unroll
        for (int j = 0; j < D; ++j) l0[j] = s_io.local[j];
#pragma unroll
        for (int j = 0; j < D; ++j) ((LdsF64*)lds)[(ExactLds<D, LPE>::Q0 + j) * WAVE + lane] = l0[j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if constexpr (!L1P) {
            loc[j] = s_io.local[j];
            qmem[j] = loc[j];
        }
        bst[j] = s_io.best[j];
    }
    s.local_cost = s_io.local_cost;
    s.best_cost = s_io.best_cost;
    s.best_sol = s_io.best_sol;
    using L = ExactLds<D, LPE>;
    static_assert(GD_ROWS(D, LPE) >= L::ROWS, "GD_ROWS (pik_kernels.hpp) must cover ExactLds");
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const int max_iters = scalar_int(max_iters_in);
    LdsF64* const lds3 = (LdsF64*)lds;
    LdsF64* const T = lds3 + lane;
    // LPE >= 4: the elite's block, and the block of this lane's line-search team (even sub-lanes q - g, odd q + g)
    LdsF64* const EB = lds3 + L::EB0 * WAVE + (lane / LPE) * L::EBS;
    LdsF64* const PF = EB + 3 * D;
    LdsF64* const TB = EB + (sub & 1) * (3 * D);
    LdsF64* const XA = EB + L::EBX;                   // frame D of the accept evaluation / its exchange slot
    LdsF64* const XT = EB + L::EBX + (sub & 1) * 12;  // exchange slot of this lane's line-search team
    // the accept evaluation's pose cost rides along with the probes when their last pass has a lane to spare;
    // the probes go one per lane (NP1 passes) or in pairs, both signs of a variable per lane (NP2 passes at ~1.45x
    // the cost of one), whichever is the shorter critical path (the accept evaluation's own pose cost, when it
    // cannot ride along, priced at 0.4 of a pass)
    constexpr bool FUSE1 = LPE >= 4 && (2 * D) % LPE != 0;
    constexpr bool FUSE2 = LPE >= 4 && D % LPE != 0;
    constexpr int NP1 = LPE >= 4 ? (2 * D + (FUSE1 ? 1 : 0) + LPE - 1) / LPE : 0;
    constexpr int NP2 = LPE >= 4 ? (D + (FUSE2 ? 1 : 0) + LPE - 1) / LPE : 0;
    constexpr bool PAIRS = LPE >= 4 && PIK_EXACT_PAIRED && 145 * NP2 + (FUSE2 ? 0 : 40) < 100 * NP1 + (FUSE1 ? 0 : 40);
    constexpr bool FUSE = PAIRS ? FUSE2 : FUSE1;
    // flat probe passes (PIK_XFLAT): 2 = one pass holds every probe and the accept evaluation -- nothing is stored in
    // front of it but the accepted point's sines / cosines; 1 = the passes start from the stored frame of their first joint
    // (class 2 in pairs, one pass: the unrolled walk of two frames with its per-joint decisions -- five kinds of origin,
    //  three axes -- takes 256 + 160 registers against 248 + 26 for the per-lane start, which it therefore keeps)
    constexpr int FLAT_ANY = (PIK_XFLAT && UZ != 0 && LPE >= 4 && D <= PIK_XTEAM_INLINE_MAXD && D <= PIK_XFLAT_MAXD)
                                 ? ((FUSE && (PAIRS ? NP2 : NP1) == 1) ? 2 : 1) : 0;
    constexpr int FLAT = (UZ == 2 && PAIRS && FLAT_ANY == 2) ? 0 : FLAT_ANY;
    (void)T;
    (void)PF;
    (void)TB;
    (void)XA;
    (void)XT;
    const int ebase = lane - sub;
    const double h = p.step_size;
    bool done = !active;
    bool first = true;
    int num_iterations = 0;
    double previous_cost = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) grd[j] = 0.0;
    s.steps = 0;
    s.iters = 0;
    s.found = 0;
    (void)ebase;
    while (__any(!done)) {
        // does some lane take a(nother) step behind this evaluation?  The iteration limit is known beforehand
        // (the data-dependent exits only ever end a descent earlier).
        const bool last = first ? (max_iters <= 0) : (MODE == GD_SINGLE || num_iterations + 1 >= max_iters);
        const int want = __any(!done && !last) ? 1 : 0;
        EvalOut e;
        if constexpr (LPE <= 2) {
            exact_accept<D, LPE, OCC, UZG>(c, p, g, seed, qmem, e, want, T, sub);
        } else if (FUSE && want) {
            wave_sync(); // (the line-search teams of the previous step have read their blocks)
            if constexpr (FLAT == 2) {
                exact_team_sincos<D, LPE>(c, qmem, EB, sub);
            } else {
                (void)exact_eval_team<D, LPE, true, FUSE ? false : true, UZG>(c, p, g, seed, qmem, EB, PF, XA, sub, 1);
                wave_sync();
            }
            if constexpr (PAIRS && FLAT && UZ == 1) {
                x_static_for<NP2>([&](auto pass) {
                    constexpr int j0 = decltype(pass)::value * LPE;
                    const CostPairSol cp = exact_probe_pair<D, LPE, UZG, FLAT, j0>(c, p, g, seed, qmem, j0, EB, PF, sub, 1);
                    const int i = j0 + sub;
                    if (i < D) {
                        lds3[(L::CM0 + i) * WAVE + ebase] = cp.a;
                        lds3[(L::CP0 + i) * WAVE + ebase] = cp.b;
                    } else if (i == D) {
                        lds3[L::AC0 * WAVE + ebase] = cp.a;
                        lds3[L::AS0 * WAVE + ebase] = cp.sol ? 1.0 : 0.0;
                    }
                });
            } else if constexpr (PAIRS) {
#pragma unroll 1
                for (int j0 = 0; j0 < D + 1; j0 += LPE) {
                    const CostPairSol cp = exact_probe_pair<D, LPE, UZG, FLAT>(c, p, g, seed, qmem, j0, EB, PF, sub, 1);
                    const int i = j0 + sub;
                    if (i < D) {
                        lds3[(L::CM0 + i) * WAVE + ebase] = cp.a;
                        lds3[(L::CP0 + i) * WAVE + ebase] = cp.b;
                    } else if (i == D) {
                        lds3[L::AC0 * WAVE + ebase] = cp.a;
                        lds3[L::AS0 * WAVE + ebase] = cp.sol ? 1.0 : 0.0;
                    }
                }
            } else {
#pragma unroll 1
                for (int probe = 0; probe < 2 * D + 1; probe += LPE) {
                    const CostSol cs = exact_probe_pass<D, LPE, UZG, FLAT>(c, p, g, seed, qmem, probe, EB, PF, sub, 1);
                    const int pr = probe + sub;
                    if (pr < 2 * D) {
                        lds3[(((pr & 1) ? L::CP0 : L::CM0) + (pr >> 1)) * WAVE + ebase] = cs.cost;
                    } else if (pr == 2 * D) {
                        lds3[L::AC0 * WAVE + ebase] = cs.cost;
                        lds3[L::AS0 * WAVE + ebase] = cs.sol ? 1.0 : 0.0;
                    }
                }
            }
            wave_sync();
            e.cost = lds3[L::AC0 * WAVE + ebase];
            e.sol = lds3[L::AS0 * WAVE + ebase] != 0.0;
        } else {
            wave_sync();
            const CostSol ce = exact_eval_team<D, LPE, true, true, UZG>(c, p, g, seed, qmem, EB, PF, XA, sub, want);
            e.cost = ce.cost;
            e.sol = ce.sol != 0;
        }
        if (first) {
            // GradientIk::from -- src/ik_gradient.cpp:14-22
            first = false;
            if (MODE != GD_SINGLE) {
                s.local_cost = e.cost;
                s.best_cost = e.cost;
            }
            s.best_sol = e.sol;
            if (!done) {
                if (MODE == GD_LOCAL && p.stop_on_valid && e.sol) {
                    s.found = 2; // ik_gradient early return, src/ik_gradient.cpp:102-104
                    done = true;
                } else if (max_iters <= 0) {
                    done = true;
                }
            }
        } else if (!done) {
            // tail of step(): always accept, update best -- src/ik_gradient.cpp:84-93
            s.local_cost = e.cost;
            s.steps += 1;
            const bool improved = e.cost < s.best_cost;
            if (improved) {
#pragma unroll
                for (int j = 0; j < D; ++j) bst[j] = L1P ? T[(L::Q0 + j) * WAVE] : loc[j];
                s.best_cost = e.cost;
                s.best_sol = e.sol;
            }
            if (MODE == GD_SINGLE) {
                done = true;
            } else if (MODE == GD_LOCAL && improved && p.stop_on_valid && e.sol) {
                s.found = 1; // src/ik_gradient.cpp:117-121
                s.iters = num_iterations + 1;
                done = true;
            } else if (fabs(e.cost - previous_cost) <= p.min_cost_delta) {
                s.iters = num_iterations;
                done = true;
            } else {
                previous_cost = e.cost;
                num_iterations += 1;
                s.iters = num_iterations;
                if (num_iterations >= max_iters) done = true;
            }
        }
        if (!__any(!done)) break;
        // head of the next step(): central differences -- src/ik_gradient.cpp:28-43
        double gr[D];
        wave_sync();
        if constexpr (L1P) {
#pragma unroll
            for (int j = 0; j < D; ++j) gr[j] = T[(L::G0 + j) * WAVE]; // (the fork left the difference)
        } else if constexpr (LPE <= 2) {
#pragma unroll
            for (int j = 0; j < D; ++j)
                gr[j] = lds3[(L::CP0 + j) * WAVE + ebase + (LPE == 2 ? 1 : 0)] - lds3[(L::CM0 + j) * WAVE + ebase];
        } else {
            if constexpr (!FUSE) { // (fused: the probes came with the accept evaluation)
                if constexpr (PAIRS && FLAT && UZ == 1) {
                    x_static_for<(D + LPE - 1) / LPE>([&](auto pass) {
                        constexpr int j0 = decltype(pass)::value * LPE;
                        const CostPairSol cp = exact_probe_pair<D, LPE, UZG, 1, j0>(c, p, g, seed, qmem, j0, EB, PF, sub, 0);
                        const int i = j0 + sub;
                        if (i < D) {
                            lds3[(L::CM0 + i) * WAVE + ebase] = cp.a;
                            lds3[(L::CP0 + i) * WAVE + ebase] = cp.b;
                        }
                    });
                } else if constexpr (PAIRS) {
#pragma unroll 1
                    for (int j0 = 0; j0 < D; j0 += LPE) {
                        const CostPairSol cp = exact_probe_pair<D, LPE, UZG, FLAT ? 1 : 0>(c, p, g, seed, qmem, j0, EB, PF, sub, 0);
                        const int i = j0 + sub;
                        if (i < D) {
                            lds3[(L::CM0 + i) * WAVE + ebase] = cp.a;
                            lds3[(L::CP0 + i) * WAVE + ebase] = cp.b;
                        }
                    }
                } else {
#pragma unroll 1
                    for (int probe = 0; probe < 2 * D; probe += LPE) {
                        const CostSol cs = exact_probe_pass<D, LPE, UZG, FLAT ? 1 : 0>(c, p, g, seed, qmem, probe, EB, PF, sub, 0);
                        const int pr = probe + sub;
                        if (pr < 2 * D) lds3[(((pr & 1) ? L::CP0 : L::CM0) + (pr >> 1)) * WAVE + ebase] = cs.cost;
                    }
                }
                wave_sync();
            }
#pragma unroll
            for (int j = 0; j < D; ++j) gr[j] = lds3[(L::CP0 + j) * WAVE + ebase] - lds3[(L::CM0 + j) * WAVE + ebase];
        }
        wave_sync();
        double p1, p3;
        double q_eval[D];
        if constexpr (L1P) {
            // normalisation -- src/ik_gradient.cpp:45-54 -- on the differences in registers; a lane that is done goes
            // through the motions on whatever its rows hold and keeps its own gradient (`grd` is only written)
            double sum = h;
#pragma unroll
            for (int j = 0; j < D; ++j) sum = sum + fabs(gr[j]);
            const double f = 1.0 / sum * h;
            double q_plus[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double gn = gr[j] * f;
                if (!done) grd[j] = gn;
                T[(L::G0 + j) * WAVE] = gn;
                if constexpr (!NG) { // (the joint goals read the two points; the walk forms them from the rows)
                    const double qj = T[(L::Q0 + j) * WAVE];
                    q_eval[j] = qj - gn;
                    q_plus[j] = qj + gn;
                } else {
                    q_eval[j] = q_plus[j] = 0.0;
                }
            }
            // line search -- src/ik_gradient.cpp:56-64
            const CostPair cp = exact_line_pair<D, OCC, UZG>(c, p, g, seed, q_eval, q_plus, T);
            p1 = cp.a;
            p3 = cp.b;
        } else {
        if (!done) {
#pragma unroll
            for (int j = 0; j < D; ++j) grd[j] = gr[j];
        }
        // normalisation -- src/ik_gradient.cpp:45-54
        double sum = h;
#pragma unroll
        for (int j = 0; j < D; ++j) sum = sum + fabs(grd[j]);
        const double f = 1.0 / sum * h;
        if (!done) {
#pragma unroll
            for (int j = 0; j < D; ++j) grd[j] = grd[j] * f;
        }
        // line search -- src/ik_gradient.cpp:56-64
        if constexpr (LPE == 1 && PIK_EXACT_PAIRED && (OCC == 1 || PIK_XPAIR_OCC2)) {
            double q_plus[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
                q_eval[j] = loc[j] - grd[j];
                q_plus[j] = loc[j] + grd[j];
            }
            const CostPair cp = exact_line_pair<D, OCC, UZG>(c, p, g, seed, q_eval, q_plus, (const LdsF64*)nullptr);
            p1 = cp.a;
            p3 = cp.b;
        } else if constexpr (LPE == 1) {
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = loc[j] - grd[j];
            evaluate<D, OCC>(c, p, g, seed, q_eval, e);
            p1 = e.cost;
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = loc[j] + grd[j];
            evaluate<D, OCC>(c, p, g, seed, q_eval, e);
            p3 = e.cost;
        } else {
            // both line probes at once: even sub-lanes q - g, odd sub-lanes q + g
            const double sg = (sub & 1) ? 1.0 : -1.0;
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = loc[j] + sg * grd[j];
            if constexpr (LPE == 2) {
                evaluate<D, OCC>(c, p, g, seed, q_eval, e);
            } else {
                const CostSol ce = exact_eval_team<D, LPE / 2, false, true, UZG>(c, p, g, seed, q_eval, TB, (LdsF64*)nullptr, XT, sub >> 1, 0);
                e.cost = ce.cost;
            }
            p1 = shfl_f64(e.cost, ebase);
            p3 = shfl_f64(e.cost, ebase + 1);
        }
        } // (!L1P)
        // secant step size + clamp -- src/ik_gradient.cpp:66-81
        const double p2 = (p1 + p3) * 0.5;
        const double cost_diff = (p3 - p1) * 0.5;
        double joint_diff = p2 / cost_diff;
        if (!isfinite(joint_diff)) joint_diff = 0.0;
        if constexpr (L1P) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double nv = clamp_joint<D>(c, j, gd_update(T[(L::Q0 + j) * WAVE], T[(L::G0 + j) * WAVE], joint_diff));
                if (!done) T[(L::Q0 + j) * WAVE] = nv;
            }
        } else if (!done) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                loc[j] = clamp_joint<D>(c, j, gd_update(loc[j], grd[j], joint_diff));
                qmem[j] = loc[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
        s_io.local[j] = L1P ? T[(L::Q0 + j) * WAVE] : loc[j];
        s_io.best[j] = bst[j];
        s_io.grad[j] = grd[j];
    }
    s_io.local_cost = s.local_cost;
    s_io.best_cost = s.best_cost;
    s_io.best_sol = s.best_sol;
    s_io.steps = s.steps;
    s_io.iters = s.iters;
    s_io.found = s.found;
}

// ---- several tip frames: the memoised descent, one or two lanes per elite -----------------------------------------
// cost_fn with several pose goals (src/goal.cpp:188-203 over the vectors of :27-49, 80-89; eval_multi, pik_math.hpp)
// is   pc = ((0 + pose_cost_0) + pose_cost_1) + ...  (+ the joint goals),   one forward kinematics per tip frame,
// and a probe q -+ h e_i of step() repeats, per tip frame,
//   * every operation of the accept evaluation's walk down that tip's path up to the frame in front of joint i
//     (the left-to-right chain product, see the head of this file), and
//   * ALL of it -- the same pose cost, bit for bit -- for a tip frame whose path does not contain variable i.
// So the accept evaluation forks the probes of variable i off its walk of every path that contains i (one sine /
// cosine and the joints from i to that tip), and the running sums of the probes take the ACCEPT evaluation's pose cost
// for the other tips -- added in the tips' order, which is the order of the literal sum.  The sines / cosines of the
// accepted point are taken once per variable instead of once per variable, tip and evaluation.  A tree of two
// five-joint arms on a common torso joint (nine variables): 90 joint-steps and 26 pose costs per step() instead of
// the literal 210 and 42, 47 sines / cosines instead of 210.  The two line-search evaluations are literal
// (`evaluate`, a call).  Chains with a floating or a mimic joint on some path keep the literal routine.
#ifndef PIK_XMULTI_MEMO
#define PIK_XMULTI_MEMO 1 // (0: several tip frames through the literal routine -- A/B experiments)
#endif
#ifndef PIK_XFLOAT_FORK
#define PIK_XFLOAT_FORK 1 // (0: a chain with a floating joint through the literal routine at every width -- A/B experiments)
#endif

// the pose-cost term of ONE tip frame and its frame tests (eval_multi's loop body behind the forward kinematics)
template <int D>
__device__ __forceinline__ double x_tip_pose_cost(CK<D> ck, PK p_in, const double* gp, const double (&R)[9],
                                                  const double (&tipt)[3], bool& ok) {
    PK p = fresh_after(p_in, tipt[0]);
    GoalK g;
    make_goal(gp, g);
    const double dx = g.t[0] - tipt[0], dy = g.t[1] - tipt[1], dz = g.t[2] - tipt[2];
    PoseErr pe;
#if PIK_XF
    pe.lin = sqrt_pos(xsumsq3(dx, dy, dz));
#else
    pe.lin = sqrt_pos(dx * dx + dy * dy + dz * dz);
#endif
    double qt[4], d0[4], vn;
    matrix_to_quat(R, qt);
    quat_mul_conj(qt, g.q, d0);
    pe.ang = angle_of(ck.mt, d0, vn);
    ok = ok && (!PIK_POS_TEST(p) || pe.lin <= p.pos_thr) && (!PIK_ORI_TEST(p) || fabs(pe.ang) <= p.ori_thr);
    return pose_cost(p, pe);
}

// the joint-goal part of cost_fn at q (eval_multi's tail): gc and, for the accept evaluation, the goal tests
template <int D>
__device__ __forceinline__ double x_joint_goals(CK<D> c, PK p, const double (&q)[D], const double (&seed)[D], bool& ok) {
    double gc = 0.0;
    if (PIK_GM(p) & 1) {
        const double w = goal_cost_term<D>(c, p, 0, q, seed) * p.w_center_sq;
        gc = gc + w;
        ok = ok && (w < p.cost_thr_sq);
    }
    if (PIK_GM(p) & 2) {
        const double w = goal_cost_term<D>(c, p, 1, q, seed) * p.w_limits_sq;
        gc = gc + w;
        ok = ok && (w < p.cost_thr_sq);
    }
    if (PIK_GM(p) & 4) {
        const double w = goal_cost_term<D>(c, p, 2, q, seed) * p.w_disp_sq;
        gc = gc + w;
        ok = ok && (w < p.cost_thr_sq);
    }
    return gc;
}

// The accept evaluation of q over every tip frame (cost + verdict, exactly eval_multi) and, when `want`, the costs of
// the probes q -+ h e_i in rows CM0 + i / CP0 + i of this lane's LDS column (LPE = 2: this lane's sign only).
template <int D, int LPE>
__device__ __noinline__ void exact_accept_multi(CK<D> c0_in, PK p_in, const GoalSet& gs, const double (&seed)[D],
                                                const double (&q)[D], EvalOut& e, int want_in, LdsF64* T, int sub) {
    static_assert(LPE <= 2, "the fork form");
    using L = ExactLds<D, LPE>;
    CK<D> c0 = scalar_ref(c0_in);
    PK p = scalar_ref(p_in);
    const int want = scalar_int(want_in);
    const int n_tips = tip_count<D>(c0);
    const double h = p.step_size;
    // the sines / cosines of the accepted point, once per variable (fk calls sincos_f64 on the same argument for every
    // tip and evaluation); a prismatic variable's are not read
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        double sn, cs;
        sincos_f64(c0.mt, q[j], sn, cs);
        T[(L::SN0 + j) * WAVE] = sn;
        T[(L::CS0 + j) * WAVE] = cs;
        T[(L::CM0 + j) * WAVE] = 0.0; // the probes' running sums over the tips
        T[(L::CP0 + j) * WAVE] = 0.0;
    }
    double pc = 0.0;
    bool ok = true;
#pragma unroll 1
    for (int k = 0; k < n_tips; ++k) { // wave-uniform trip count
        CK<D> ck = tip_chain<D>(c0, k);
        const double* const gp = gs.ptr + 7 * k;
        const uint32_t active = ck.active_mask, pris = ck.prismatic_mask, kinds = ck.axis_kind;
        double R[9], t[3];
        R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
        R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
        R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
        t[0] = t[1] = t[2] = 0.0;
#pragma unroll 1
        for (int j = 0; j < D; ++j) {
            if (!((active >> j) & 1u)) continue; // not a joint of this tip's path
            chain_origin<D>(ck, j, R, t, false); // (fk<MASKED>: nothing is copied, the first origin multiplies the identity)
            const bool pj = (pris >> j) & 1u;
            const uint32_t kj = (kinds >> (2 * j)) & 3u;
            const double qj = q[j];
            if (want) {
                // the probes of variable j branch off here: (R, t) is this path's frame in front of joint j
                constexpr int NS = LPE == 2 ? 1 : 2;
#pragma unroll 1
                for (int it = 0; it < NS; ++it) {
                    const int sg = LPE == 2 ? (sub & 1) : it;
                    const double vj = qj + (sg ? h : -h);
                    double R2[9], t2[3];
#pragma unroll
                    for (int i = 0; i < 9; ++i) R2[i] = R[i];
                    t2[0] = t[0];
                    t2[1] = t[1];
                    t2[2] = t[2];
                    double sn = 0.0, cs = 1.0;
                    if (!pj) sincos_f64(ck.mt, vj, sn, cs);
                    chain_joint<D>(ck, j, R2, t2, pj, kj, vj, sn, cs);
#pragma unroll 1
                    for (int m = j + 1; m < D; ++m) {
                        if (!((active >> m) & 1u)) continue;
                        chain_origin<D>(ck, m, R2, t2, false);
                        chain_joint<D>(ck, m, R2, t2, (pris >> m) & 1u, (kinds >> (2 * m)) & 3u, q[m],
                                       T[(L::SN0 + m) * WAVE], T[(L::CS0 + m) * WAVE]);
                    }
                    if (!ck.tip_ident) iso_mul(R2, t2, ck.tip);
                    bool ok2 = true;
                    const double pcp = x_tip_pose_cost<D>(ck, p, gp, R2, t2, ok2);
                    const int row = (sg ? L::CP0 : L::CM0) + j;
                    T[row * WAVE] = T[row * WAVE] + pcp;
                }
            }
            chain_joint<D>(ck, j, R, t, pj, kj, qj, T[(L::SN0 + j) * WAVE], T[(L::CS0 + j) * WAVE]);
        }
        if (!ck.tip_ident) iso_mul(R, t, ck.tip);
        const double pck = x_tip_pose_cost<D>(ck, p, gp, R, t, ok);
        pc = pc + pck;
        if (want) {
            // a probe of a variable that is not on this path moves nothing of it: the accept evaluation's term
#pragma unroll 1
            for (int j = 0; j < D; ++j) {
                if ((active >> j) & 1u) continue;
                T[(L::CM0 + j) * WAVE] = T[(L::CM0 + j) * WAVE] + pck;
                T[(L::CP0 + j) * WAVE] = T[(L::CP0 + j) * WAVE] + pck;
            }
        }
    }
    PK pg = fresh_after(p, pc);
    CK<D> cg = fresh_after(c0, pc);
    double cost = pc;
    if (PIK_GM(pg)) {
        cost = cost + x_joint_goals<D>(cg, pg, q, seed, ok);
        if (want) {
            // the joint goals of the probes: the literal sums at q -+ h e_j
#pragma unroll 1
            for (int j = 0; j < D; ++j) {
                constexpr int NS = LPE == 2 ? 1 : 2;
#pragma unroll 1
                for (int it = 0; it < NS; ++it) {
                    const int sg = LPE == 2 ? (sub & 1) : it;
                    const double dh = sg ? h : -h;
                    double qp[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) qp[i] = q[i] + ((i == j) ? dh : 0.0);
                    bool ok2 = true;
                    const int row = (sg ? L::CP0 : L::CM0) + j;
                    T[row * WAVE] = T[row * WAVE] + x_joint_goals<D>(cg, pg, qp, seed, ok2);
                }
            }
        }
    }
    e.cost = cost;
    e.pc = pc;
    e.lin = e.ang = e.vn = 0.0;
    e.g0 = e.g1 = e.g2 = 0.0;
    e.sol = ok;
}


// ---- one tip frame behind a FLOATING joint (seven variables, ONE transform Translation(t) * Quaterniond(w, x, y, z) at
// the joint's seventh variable; fk's PIK_STRICT path, MoveIt's FloatingJointModel::computeTransform) ------------------
// The literal routine evaluates every probe from scratch through a called `evaluate` -- for the Panda on a free-flying
// base (fourteen variables) 31 calls per step(), 217 sines / cosines, and the stack frames of the calls were 1.35 MB of
// HBM traffic per solved problem.  The fork form: the probes of the floating joint's seven variables all branch off in
// front of its transform (the frame behind its origin; the transform itself rebuilt from the perturbed variable), the
// probes of every other joint where the walk stands in front of that joint, and the joints behind take the accept
// evaluation's sines / cosines and the accept evaluation's floating transform -- the same operations on the same
// operands as the literal evaluation of the probe.  ONE floating joint, no mimic joint (x_float_fork_ok); one or two
// lanes per elite (the wider kernels keep the literal routine, whose probes are dealt out to the lanes).
// LDS rows of the lane's column beyond ExactLds<D, LPE <= 2>'s 4 D: the accepted point itself (rows XFQ0 + j), so that
// the variables of the floating joint are read by a dynamic index without a private array behind a pointer.
template <int D>
struct ExactFloatLds {
    static constexpr int XFQ0 = 4 * D;
    static constexpr int ROWS = 5 * D;
};
// the floating joint's transform from its seven values v = (tx ty tz rx ry rz rw), as fk builds it
__device__ __forceinline__ void x_float_transform(const double (&v)[7], double (&J)[12]) {
    const double qq[4] = {v[6], v[3], v[4], v[5]};
    double JR[9];
    quat_to_matrix(qq, JR);
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = JR[i];
    J[9] = v[0];
    J[10] = v[1];
    J[11] = v[2];
}
template <int D, int LPE>
__device__ __noinline__ void exact_accept_float(CK<D> c_in, PK p_in, const GoalK& g_in, const double (&seed)[D],
                                                const double (&q)[D], EvalOut& e, int want_in, LdsF64* T, int sub) {
    static_assert(LPE <= 2, "the fork form");
    using L = ExactLds<D, LPE>;
    using F = ExactFloatLds<D>;
    static_assert(GD_ROWS(D, LPE) >= F::ROWS, "GD_ROWS (pik_kernels.hpp) must cover ExactFloatLds");
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const GoalK g = g_in;
    const int want = scalar_int(want_in);
    const uint32_t pris = c.prismatic_mask, kinds = c.axis_kind, skip_mask = c.skip_mask;
    const int jf = __builtin_ctz(c.float_mask); // the floating joint's seventh variable (wave-uniform)
    const double h = p.step_size;
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        double sn, cs;
        sincos_f64(c.mt, q[j], sn, cs); // (read for the revolute joints only)
        T[(L::SN0 + j) * WAVE] = sn;
        T[(L::CS0 + j) * WAVE] = cs;
        T[(F::XFQ0 + j) * WAVE] = q[j];
    }
    double fv[7], J0[12];
#pragma unroll
    for (int i = 0; i < 7; ++i) fv[i] = T[(F::XFQ0 + jf - 6 + i) * WAVE];
    x_float_transform(fv, J0);
    // (R2, t2) through the joints m0 .. D - 1 at the accepted point, then the tip transform
    auto rest_walk = [&](double (&R2)[9], double (&t2)[3], int m0) {
#pragma unroll 1
        for (int m = m0; m < D; ++m) {
            if ((skip_mask >> m) & 1u) continue;
            chain_origin<D>(c, m, R2, t2, false);
            if (m == jf) iso_mul_r(R2, t2, J0);
            else chain_joint<D>(c, m, R2, t2, (pris >> m) & 1u, (kinds >> (2 * m)) & 3u, T[(F::XFQ0 + m) * WAVE],
                                T[(L::SN0 + m) * WAVE], T[(L::CS0 + m) * WAVE]);
        }
        if (!c.tip_ident) iso_mul(R2, t2, c.tip);
    };
    double R[9], t[3];
    R[0] = 1.0; R[1] = 0.0; R[2] = 0.0;
    R[3] = 0.0; R[4] = 1.0; R[5] = 0.0;
    R[6] = 0.0; R[7] = 0.0; R[8] = 1.0;
    t[0] = t[1] = t[2] = 0.0;
    bool blank = true;
    constexpr int NS = LPE == 2 ? 1 : 2;
#pragma unroll 1
    for (int j = 0; j < D; ++j) {
        if ((skip_mask >> j) & 1u) continue; // one of the floating joint's first six variables: no transform of its own
        chain_origin<D>(c, j, R, t, blank);
        const bool pj = (pris >> j) & 1u;
        const uint32_t kj = (kinds >> (2 * j)) & 3u;
        const double qj = T[(F::XFQ0 + j) * WAVE];
        if (want) {
            // the probes of variable j -- of the seven variables jf - 6 .. jf when j is the floating joint -- branch off
            // here: (R, t) is the frame in front of joint j's own transform
            const int nv = (j == jf) ? 7 : 1;
#pragma unroll 1
            for (int v = 0; v < nv; ++v) {
#pragma unroll 1
                for (int it = 0; it < NS; ++it) {
                    const int sg = LPE == 2 ? (sub & 1) : it;
                    const double dh = sg ? h : -h;
                    double R2[9], t2[3];
#pragma unroll
                    for (int i = 0; i < 9; ++i) R2[i] = R[i];
                    t2[0] = t[0];
                    t2[1] = t[1];
                    t2[2] = t[2];
                    int var = j;
                    if (j == jf) {
                        var = jf - 6 + v;
                        double pv[7], Jp[12];
#pragma unroll
                        for (int i = 0; i < 7; ++i) pv[i] = (i == v) ? fv[i] + dh : fv[i];
                        x_float_transform(pv, Jp);
                        iso_mul_r(R2, t2, Jp);
                    } else {
                        const double vj = qj + dh;
                        double sn = 0.0, cs = 1.0;
                        if (!pj) sincos_f64(c.mt, vj, sn, cs);
                        chain_joint<D>(c, j, R2, t2, pj, kj, vj, sn, cs);
                    }
                    rest_walk(R2, t2, j + 1);
                    EvalOut e2;
                    double d2[4];
                    pose_tail<D, true, false>(c, p, g, seed, q, R2, t2, e2, d2, var, dh);
                    T[((sg ? L::CP0 : L::CM0) + var) * WAVE] = e2.cost;
                }
            }
        }
        if (j == jf) iso_mul_r(R, t, J0);
        else chain_joint<D>(c, j, R, t, pj, kj, qj, T[(L::SN0 + j) * WAVE], T[(L::CS0 + j) * WAVE]);
        blank = false;
    }
    if (!c.tip_ident) iso_mul(R, t, c.tip);
    double d0[4];
    pose_tail<D, false, false>(c, p, g, seed, q, R, t, e, d0);
}

// exactly one floating joint and no mimic joint? (wave-uniform; everything else keeps the literal routine)
template <int D>
__device__ __forceinline__ bool x_float_fork_ok(CK<D> c) {
    return c.float_mask != 0u && (c.float_mask & (c.float_mask - 1u)) == 0u && c.m_count == 0u;
}

// step() and its driver loops around a fork-form accept evaluation that has no specialised descent of its own -- several
// tip frames (G = GoalSet: exact_accept_multi) or one tip frame behind a floating joint (G = GoalK: exact_accept_float)
// -- with the bookkeeping of gradient_descent_exact at LPE <= 2 and the state in the caller's frame; the two line-search
// evaluations are literal (`evaluate`)
template <int D, int MODE, int LPE, typename G>
__device__ __noinline__ void gradient_descent_exact_fork(CK<D> c_in, PK p_in, const G& gs, const double (&seed)[D],
                                                         GdState<D>& s, bool active, int max_iters_in, double* lds,
                                                         int lane, int sub) {
    static_assert(LPE <= 2, "the fork form: one lane per elite, or two");
    using L = ExactLds<D, LPE>;
    static_assert(GD_ROWS(D, LPE) >= L::ROWS, "GD_ROWS (pik_kernels.hpp) must cover ExactLds");
    CK<D> c = scalar_ref(c_in);
    PK p = scalar_ref(p_in);
    const int max_iters = scalar_int(max_iters_in);
    LdsF64* const lds3 = (LdsF64*)lds;
    LdsF64* const T = lds3 + lane;
    const int ebase = lane - sub;
    const double h = p.step_size;
    bool done = !active;
    bool first = true;
    int num_iterations = 0;
    double previous_cost = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) s.grad[j] = 0.0;
    s.steps = 0;
    s.iters = 0;
    s.found = 0;
    while (__any(!done)) {
        const bool last = first ? (max_iters <= 0) : (MODE == GD_SINGLE || num_iterations + 1 >= max_iters);
        const int want = __any(!done && !last) ? 1 : 0;
        EvalOut e;
        if constexpr (std::is_same<G, GoalSet>::value) exact_accept_multi<D, LPE>(c, p, gs, seed, s.local, e, want, T, sub);
        else exact_accept_float<D, LPE>(c, p, gs, seed, s.local, e, want, T, sub);
        if (first) {
            // GradientIk::from -- src/ik_gradient.cpp:14-22
            first = false;
            if (MODE != GD_SINGLE) {
                s.local_cost = e.cost;
                s.best_cost = e.cost;
            }
            s.best_sol = e.sol;
            if (!done) {
                if (MODE == GD_LOCAL && p.stop_on_valid && e.sol) {
                    s.found = 2; // ik_gradient early return, src/ik_gradient.cpp:102-104
                    done = true;
                } else if (max_iters <= 0) {
                    done = true;
                }
            }
        } else if (!done) {
            // tail of step(): always accept, update best -- src/ik_gradient.cpp:84-93
            s.local_cost = e.cost;
            s.steps += 1;
            const bool improved = e.cost < s.best_cost;
            if (improved) {
#pragma unroll
                for (int j = 0; j < D; ++j) s.best[j] = s.local[j];
                s.best_cost = e.cost;
                s.best_sol = e.sol;
            }
            if (MODE == GD_SINGLE) {
                done = true;
            } else if (MODE == GD_LOCAL && improved && p.stop_on_valid && e.sol) {
                s.found = 1; // src/ik_gradient.cpp:117-121
                s.iters = num_iterations + 1;
                done = true;
            } else if (fabs(e.cost - previous_cost) <= p.min_cost_delta) {
                s.iters = num_iterations;
                done = true;
            } else {
                previous_cost = e.cost;
                num_iterations += 1;
                s.iters = num_iterations;
                if (num_iterations >= max_iters) done = true;
            }
        }
        if (!__any(!done)) break;
        // head of the next step(): central differences -- src/ik_gradient.cpp:28-43
        double gr[D];
        wave_sync();
#pragma unroll
        for (int j = 0; j < D; ++j)
            gr[j] = lds3[(L::CP0 + j) * WAVE + ebase + (LPE == 2 ? 1 : 0)] - lds3[(L::CM0 + j) * WAVE + ebase];
        wave_sync();
        if (!done) {
#pragma unroll
            for (int j = 0; j < D; ++j) s.grad[j] = gr[j];
        }
        // normalisation -- src/ik_gradient.cpp:45-54
        double sum = h;
#pragma unroll
        for (int j = 0; j < D; ++j) sum = sum + fabs(s.grad[j]);
        const double f = 1.0 / sum * h;
        if (!done) {
#pragma unroll
            for (int j = 0; j < D; ++j) s.grad[j] = s.grad[j] * f;
        }
        // line search -- src/ik_gradient.cpp:56-64
        double p1, p3;
        double q_eval[D];
        if constexpr (LPE == 1) {
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] - s.grad[j];
            evaluate<D>(c, p, gs, seed, q_eval, e);
            p1 = e.cost;
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + s.grad[j];
            evaluate<D>(c, p, gs, seed, q_eval, e);
            p3 = e.cost;
        } else {
            const double sg = (sub & 1) ? 1.0 : -1.0;
#pragma unroll
            for (int j = 0; j < D; ++j) q_eval[j] = s.local[j] + sg * s.grad[j];
            evaluate<D>(c, p, gs, seed, q_eval, e);
            p1 = shfl_f64(e.cost, ebase);
            p3 = shfl_f64(e.cost, ebase + 1);
        }
        // secant step size + clamp -- src/ik_gradient.cpp:66-81
        const double p2 = (p1 + p3) * 0.5;
        const double cost_diff = (p3 - p1) * 0.5;
        double joint_diff = p2 / cost_diff;
        if (!isfinite(joint_diff)) joint_diff = 0.0;
        if (!done) {
            CK<D> cl = fresh_after(c, joint_diff);
#pragma unroll
            for (int j = 0; j < D; ++j) s.local[j] = clamp_joint<D>(cl, j, gd_update(s.local[j], s.grad[j], joint_diff));
        }
    }
}

// no floating / mimic joint on any tip's path? (wave-uniform; those keep the literal routine)
template <int D>
__device__ __forceinline__ bool x_multi_memo_ok(CK<D> c0) {
    const int n_tips = tip_count<D>(c0);
    uint32_t bad = 0u;
    for (int k = 0; k < n_tips; ++k) {
        CK<D> ck = tip_chain<D>(c0, k);
        bad |= ck.float_mask | ck.skip_mask | ck.m_count;
    }
    return bad == 0u;
}

} // namespace pik

#endif // PIK_STRICT
