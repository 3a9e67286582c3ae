#!/usr/bin/env python3
"""EXPERIMENT (round 4, not shipped): generates a variant of streamchat_amd/csrc/gemm.hip in which iteration 0 of a workgroup's NEXT tile runs
inside the epilogue of the current one (k_gemm_fat<EPI, PERSIST, EITER>, plain / quick-GELU / erf-GELU epilogues without a residual).
    python tools/experiments/gemm_eiter_patch.py   ->  /tmp/eiter/gemm_eiter.hip
What it showed (profiles/r04_run11_gemm_eiter_experiment.md): hipcc cannot hold the merged epilogue in 256 VGPRs - the four fragment sets the
MFMAs need (a0 / b0 / b1 + a double-buffered a1: 104 registers) on top of the epilogue's own values push the VGPR / AGPR split past 256 and it
spills ACCUMULATOR tiles to scratch (94 scratch loads, 52 stores in the NONE instantiation), each behind an s_waitcnt vmcnt(0) that drains the
DMA queue.  The merged epilogue would have to be register-allocated by hand like the K loop."""
import os, re, sys
os.makedirs('/tmp/eiter', exist_ok=True)
s=open('/root/repo/streamchat_amd/csrc/gemm.hip').read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)>=1, old[:80]
    s=s.replace(old,new,cnt)
# A. template signature
rep('''template <int EPI, bool PERSIST>
__global__ __launch_bounds__(256, 1) void k_gemm_fat(''','''template <int EPI, bool PERSIST, bool EITER = false>
__global__ __launch_bounds__(256, 1) void k_gemm_fat(''')
# B. loop restructure: tile_top lambda + peeled iteration 0
old_top=s[s.index('    for (;;) {                                                  // tiles of this workgroup (one unless PERSIST)\n'):s.index('    FAT_STAMP(0);\n    iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);')]
body=old_top[old_top.index('    {\n        const int n0b'):]   # the bias/residual DMA block
new_top='''    // EITER (round 4): iteration 0 of a workgroup's NEXT tile runs inside the epilogue of the current one (its fragments are in registers
    // and LDS already; an accumulator tile is free as soon as the epilogue has read it), so the tile loop starts at iteration 1 and
    // iteration 0 is peeled in front of it for the workgroup's first tile.  Without EITER iteration 0 follows the epilogue as a whole.
    auto tile_top = [&]() {
''' + body.replace('\n    {\n','\n',1) if False else None
rep(old_top, '''    auto tile_top = [&]() {
''' + body + '''    };
    tile_top();
    iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);    // (writes every accumulator: nothing to zero)
    for (;;) {                                                  // tiles of this workgroup (one unless PERSIST)
''')
rep('''    FAT_STAMP(0);
    iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);    // (writes every accumulator: nothing to zero)
    iter(std::integral_constant<int, 1>{}, std::false_type{}, 1);''','''    FAT_STAMP(0);
    iter(std::integral_constant<int, 1>{}, std::false_type{}, 1);''')
# end of loop: after advance, tile_top + iter0 (non-EITER)
rep('''    if (has_nx) { tile_of(vb + (int)gridDim.x, tm_nx, tn_nx); tile_src(tm_nx, tn_nx, At_nx, Wt_nx, a_ext_nx, w_ext_nx); }
    }
    // The last iteration of the last tile still issued''','''    if (has_nx) { tile_of(vb + (int)gridDim.x, tm_nx, tn_nx); tile_src(tm_nx, tn_nx, At_nx, Wt_nx, a_ext_nx, w_ext_nx); }
    tile_top();
    if (!EITER) iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);      // (EITER: done inside the epilogue above)
    }
    // The last iteration of the last tile still issued''')
# C. E-iter plain epilogue
old_plain='''            sc_u2 rz[8] = {};
            sc_u4 d[4];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                cmath(mi, bv, rz, false);
                if (mi > 0) cstore(mi - 1, d);
                cslab_rd(d);
            }
            cstore(7, d);'''
new_plain='''            sc_u2 rz[8] = {};
            sc_u4 d[4];
            if (EITER && has_nx) {
                // ---- the epilogue with the next tile's iteration 0 inside it.  In this tile's DMA frame that iteration is k = nk: its operands are
                // in buffer 0 (fetched by iteration nk - 2), its K-half-0 fragments in a0 / b0 (read at RC of iteration nk - 1).  Every duty of
                // iter(X = 0, FIRST, k = nk) happens here: the K-half-1 fragments (all of W's up front, A's one row tile ahead of its use: only
                // two of them live), the barrier that frees buffer 0's W planes and K-half-0 A plane, 12 of the 16 DMA rounds of iteration nk + 2
                // spread over the row tiles (the 4 rounds into the K-half-1 A plane follow the last fragment read), the 128 MFMAs - two per
                // column step, into accumulator tiles the math of that step has just read, K half 1 right behind K half 0 of the previous
                // column - and at the end the wait for iteration nk + 1's operands + the fragment reads of its K half 0.  The next tile then
                // starts at its iteration 1.  (sched_barriers keep each column step's math between its MFMAs: the wave issues in order, a
                // burst of MFMAs hides nothing.)
#define FAT_RD8(dst, ad) FAT_RD(dst[0], ad, 0); FAT_RD(dst[1], ad, 1); FAT_RD(dst[2], ad, 2); FAT_RD(dst[3], ad, 3); FAT_RD(dst[4], ad, 4); FAT_RD(dst[5], ad, 5); FAT_RD(dst[6], ad, 6); FAT_RD(dst[7], ad, 7)
                sc_u2 bq[8];
#pragma unroll
                for (int nj = 0; nj < 8; ++nj) bq[nj] = *reinterpret_cast<const sc_u2*>(bslot + nj * 32);
                FAT_RD8(b1, b_ad[1][0]);
                FAT_RD(a1[0], a_ad[1][0], 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                constexpr int early[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14};         // W rounds, then the K-half-0 rounds of A
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (mi + 1 < 8) FAT_RD(a1[(mi + 1) & 1], a_ad[1][0], mi + 1);
#pragma unroll
                    for (int nj = 0; nj < 8; ++nj) {
                        float v[4], b4[4];
                        h4f(bq[nj], b4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = EPI == SC_EPI_COLSCALE ? (acc[mi][nj][e] + b4[e]) * cscale : epi_apply(acc[mi][nj][e] + b4[e], EPI);
                        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));          // the accumulator tile has been READ: it may be overwritten now
                        FAT_MM0(mi, nj, a0, b0);
                        if (nj > 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[mi][nj - 1]) : "v"(b1[nj - 1]), "v"(a1[mi & 1]));
                        *reinterpret_cast<sc_u2*>(cslab + (acc_o ^ (nj * 32))) = sc_u2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[mi][7]) : "v"(b1[7]), "v"(a1[mi & 1]));
                    dma(early[mi], 0, nk + 2);
                    if (mi < 4) dma(early[8 + mi], 0, nk + 2);
                    if (mi > 0) cstore(mi - 1, d);
                    cslab_rd(d);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              // the slab rows of this step and the A fragment of the next one
                }
                cstore(7, d);
                __builtin_amdgcn_s_barrier();                                                       // every wave has read its last K-half-1 A fragment
                dma(9, 0, nk + 2); dma(11, 0, nk + 2); dma(13, 0, nk + 2); dma(15, 0, nk + 2);
                // iteration nk + 1 (fetched by iteration nk - 1, before this epilogue) has landed when at most this epilogue's own 32 C stores and 16
                // DMA rounds are still in flight
                asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                FAT_RD8(b0, b_ad[0][1]);
                FAT_RD8(a0, a_ad[0][1]);
#undef FAT_RD8
            } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                cmath(mi, bv, rz, false);
                if (mi > 0) cstore(mi - 1, d);
                cslab_rd(d);
            }
            cstore(7, d);
            }'''
rep(old_plain,new_plain)
# residual path unreachable under EITER
rep('''        } else if (R) {
            // (the bias values are made opaque''','''        } else if (!EITER && R) {
            // (the bias values are made opaque''')
# D. launcher
rep('''            const bool fp = persist && fat != 2 && nt_all > n_cu;
            static bool fattr[16][8][2] = {};''','''            const bool fp = persist && fat != 2 && nt_all > n_cu;
            static int eiter_on = -1;
            if (eiter_on < 0) { const char* e = getenv("SC_GEMM_EITER"); eiter_on = e ? atoi(e) : 1; }
            if (eiter_on && fp && !R && K >= 256 && (EPI == SC_EPI_NONE || EPI == SC_EPI_QUICK_GELU || EPI == SC_EPI_GELU_ERF)) {
                static bool eattr[16][8] = {};
                if (!eattr[dev][EPI]) { (void)hipFuncSetAttribute((const void*)k_gemm_fat<EPI, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); eattr[dev][EPI] = true; }
                hipLaunchKernelGGL((k_gemm_fat<EPI, true, true>), dim3(n_cu), dim3(256), 163840, s, (const _Float16*)A, lda, (const _Float16*)W, (const _Float16*)bias,
                                   (const _Float16*)nullptr, 0, C, ldc, M, N, K, tN, gm_sel, nt_all, (const float*)nullptr, 0, 0, 1.0f);
                SC_CHECK_LAUNCH("sc_gemm_f16");
                return SC_OK;
            }
            static bool fattr[16][8][2] = {};''')
open('/tmp/eiter/gemm_eiter.hip','w').write(s)
print("ok")
