// One plane iteration of the two-step excited kernel (wafer_stencil_x2.hip.h, wafer_k_xstep2): included as TEXT into the plane
// loop -- three times, with WAFER_X2_PH = 0, 1, 2, where the z-queues are rings (RING; see wafer_stencil_fused3_iter.inc.h).
// `it` is the iteration; everything else is the enclosing kernel's.  The Y1 queue advances in mid-iteration: WAFER_X2_Q1.
        const int z = z1 + it;
        const bool more = it + 1 < niter;
#if WAFER_DIAG & 2    // timing experiment: every prefetch asks for the column's first planes again (cache hits)
        const long long zo = (long long)(z1 + (it & 1)) * g.plane;
#else
        const long long zo = (long long)z * g.plane;
#endif
        // ---- 1. prefetch, raw: input and stored states two planes ahead, V one plane ahead
        //         (in the storage type: what arrives is widened in step 5, behind the barrier)
        SVT pre[RY], pre_l[NL][RY], pre_m[NL][RY], pre_v[RY], xpre = szero, xpre_l[NL], xpre_m[NL], xpre_v = szero;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pre[r] = pre_v[r] = szero;
#pragma unroll
            for (int j = 0; j < NL; ++j) pre_l[j][r] = pre_m[j][r] = szero;
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) xpre_l[j] = xpre_m[j] = szero;
        // The requests of a wave are spread over the iteration (as in the three-step kernel: all eight waves leave the barrier at
        // once, and (2 + 4k) x 8 requests of 1 KiB queueing at the CU's one address unit kept every wave from its arithmetic):
        // which group goes where was measured per tile shape (profiles/r04_ab_x2_request_placement.jsonl): A (the main rows' input and V)
        // and L (their stored states) at the top, M (the images M_j) behind level 1 of the extra slot on the tall tile and at the top on the
        // low one, X (the extra slot's) behind level 2.
        auto issue_group = [&](int pos) {
            if (pos == 0) {   // A
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    pre[r] = *reinterpret_cast<const SVT *>((phi + zo + 2 * g.plane + rowoff[r]) + xlu);
                    if constexpr (VG == 0) pre_v[r] = *reinterpret_cast<const SVT *>((pv + zo + g.plane + rowoff[r]) + xlu);
                }
            }
            if (pos == 0) {   // L
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int j = 0; j < NL; ++j) pre_l[j][r] = *reinterpret_cast<const SVT *>((WAFER_X2_L(j) + zo + 2 * g.plane + rowoff[r]) + xlu);
            }
            if (pos == (RY == 2 ? 2 : 0)) {   // M
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int j = 0; j < NL; ++j) pre_m[j][r] = *reinterpret_cast<const SVT *>((WAFER_X2_M(j) + zo + 2 * g.plane + rowoff[r]) + xlu);
            }
            if (pos == 3) {   // X
                if (x_row) {
                    xpre = *reinterpret_cast<const SVT *>((phi + zo + 2 * g.plane + xoff_row) + xlu);
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        xpre_l[j] = *reinterpret_cast<const SVT *>((WAFER_X2_L(j) + zo + 2 * g.plane + xoff_row) + xlu);
                        xpre_m[j] = *reinterpret_cast<const SVT *>((WAFER_X2_M(j) + zo + 2 * g.plane + xoff_row) + xlu);
                    }
                    // (every row wave, also the two that only stage their row: a request inside one more branch makes the wait-count pass
                    //  wait for the requests issued before it)
                    if constexpr (VG == 0) xpre_v = *reinterpret_cast<const SVT *>((pv + zo + g.plane + xoff_row) + xlu);
                } else {
                    xpre[0] = phi[zo + 2 * g.plane + c_off];
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        xpre_l[j][0] = WAFER_X2_L(j)[zo + 2 * g.plane + c_off];
                        xpre_m[j][0] = WAFER_X2_M(j)[zo + 2 * g.plane + c_off];
                    }
                    if constexpr (VG == 0) xpre_v[0] = pv[zo + g.plane + c_off];
                }
            }
        };
        issue_group(0);
        // ---- 2. stage the next x0 plane into the other buffer
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
#pragma unroll
            for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 2)) * LP0 + HX0 + xl) = q0[WAFER_X2_Q0(2)][r];
            if (x_row) *reinterpret_cast<VT *>(nt + (xy - (y0 - 2)) * LP0 + HX0 + xl) = xy_out ? zero : xq0[WAFER_X2_Q0(2)];
            else if (c_ok) nt[c_lds0] = c_xout ? T(0) : xq0[WAFER_X2_Q0(2)][0];
        }
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        T *w1 = lds1 + (z & 1) * Cfg::TILE1;
        const T *c1 = lds1 + ((z + 1) & 1) * Cfg::TILE1;
        const bool wplane1 = work_plane(z), wplane2 = work_plane(z - 1);
        const int zp2 = z - 1;
        const bool act2 = zp2 >= zs && zp2 < ze;
        VT p1new[RY], canew[RY], cbnew[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = canew[r] = cbnew[r] = zero;
        VT xp1 = zero;

        bool all_rows = x0 + TX <= g.nx;
#pragma unroll
        for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];
        // ---- 2b. x / y neighbours of the main rows at both levels, requested ahead of the arithmetic (what level L reads
        //          this iteration was written before the last barrier; the slots written this iteration are the other ones)
        T nbl[2][RY], nbr[2][RY];
        VT nbu[2], nbd[2];
        auto nbload = [&](auto level_tag) {
            constexpr int L = decltype(level_tag)::value;
            const T *cc = L == 0 ? c0 : c1;
            constexpr int lp = L == 0 ? LP0 : LP1, hx = L == 0 ? HX0 : HX1;
            const int yb = y0 - 2 + L;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                const int o = (yrow[r] - yb) * lp + hx + xl;
                nbl[L][r] = cc[o - 1];
                nbr[L][r] = cc[o + VEC];
            }
            nbu[L] = *reinterpret_cast<const VT *>(cc + (yrow[0] - yb - 1) * lp + hx + xl);
            nbd[L] = *reinterpret_cast<const VT *>(cc + (yrow[RY - 1] - yb + 1) * lp + hx + xl);
        };
        nbload(std::integral_constant<int, 0>{});
        // ---- 3. level 1, main rows
        auto level1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT res = zero;
                if (INTERIOR || (wplane1 && rowwk[r])) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const C w = (C)q0[WAFER_X2_Q0(1)][r][v];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)q0[WAFER_X2_Q0(0)][r][v]; zz[1] = w; zz[2] = (C)q0[WAFER_X2_Q0(2)][r][v];
                        xs[1] = ys[1] = w;
                        xs[0] = (v >= 1) ? (C)q0[WAFER_X2_Q0(1)][r][(v + VEC - 1) % VEC] : (C)nbl[0][r];
                        xs[2] = (v + 1 < VEC) ? (C)q0[WAFER_X2_Q0(1)][r][(v + 1) % VEC] : (C)nbr[0][r];
                        ys[0] = (r >= 1) ? (C)q0[WAFER_X2_Q0(1)][r >= 1 ? r - 1 : 0][v] : (C)nbu[0][v];
                        ys[2] = (r + 1 < RY) ? (C)q0[WAFER_X2_Q0(1)][r + 1 < RY ? r + 1 : RY - 1][v] : (C)nbd[0][v];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        C ka, kb;
                        const T rs = update_keep(w, v_at((C)vcur[r][v], xi + v, yrow[r], z), S, ka, kb);
                        canew[r][v] = (T)ka;
                        cbnew[r][v] = (T)kb;
                        res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                    }
                }
                p1new[r] = res;
                *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - 1)) * LP1 + HX1 + xl) = res;
            }
        };
        if (all_rows && wplane1) level1(std::true_type{});
        else level1(std::false_type{});
        issue_group(1);
        // level 2's neighbours and the stored states of the plane it is about to produce (the lane's own LDS queue), requested
        // behind level 1's arithmetic
        VT lq[NL][RY];
        if (XS || act2) {
            nbload(std::integral_constant<int, 1>{});
            const QT *qs = qslot(z - 1);
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int r = 0; r < RY; ++r) lq[j][r] = wafer_f3_widen<QVT, VT, 2>(*reinterpret_cast<const QVT *>(qs + j * QS + qoff[r]));
        }
        // ---- 3x. level 1, the extra slot
        if (x_row) {
            if (x_l1) {
                VT res = zero;
                if (wplane1 && xwk) {
                    const int ly = xy - (y0 - 2);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const C w = (C)xq0[WAFER_X2_Q0(1)][v];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)xq0[WAFER_X2_Q0(0)][v]; zz[1] = w; zz[2] = (C)xq0[WAFER_X2_Q0(2)][v];
                        xs[1] = ys[1] = w;
                        xs[0] = (v >= 1) ? (C)xq0[WAFER_X2_Q0(1)][(v + VEC - 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v - 1];
                        xs[2] = (v + 1 < VEC) ? (C)xq0[WAFER_X2_Q0(1)][(v + 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v + 1];
                        ys[0] = (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                        ys[2] = (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        C ka, kb;
                        const T rs = update_keep(w, v_at((C)xv[v], xi + v, xy, z), S, ka, kb);
                        res[v] = (xi + v < g.nx) ? rs : T(0);
                    }
                }
                xp1 = res;
                *reinterpret_cast<VT *>(w1 + (xy - (y0 - 1)) * LP1 + HX1 + xl) = res;
            }
        } else if (c_l1) {
            T rs = T(0);
            if (wplane1 && c_wk) {
                const C w = (C)xq0[WAFER_X2_Q0(1)][0];
                C xs[3], ys[3], zz[3];
                zz[0] = (C)xq0[WAFER_X2_Q0(0)][0]; zz[1] = w; zz[2] = (C)xq0[WAFER_X2_Q0(2)][0];
                xs[1] = ys[1] = w;
                xs[0] = (C)c0[c_lds0 - 1]; xs[2] = (C)c0[c_lds0 + 1];
                ys[0] = (C)c0[c_lds0 - LP0]; ys[2] = (C)c0[c_lds0 + LP0];
                const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                C ka, kb;
                rs = update_keep(w, v_at((C)xv[0], cxw, cy, z), S, ka, kb);
            }
            w1[c_lds1] = rs;
            xp1[0] = rs;
        }
        (void)xp1;
        issue_group(2);
        // ---- 4. level 2: Z of the plane behind from the Y1 queue, a, b as level 1 formed them one iteration ago; the sums
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if constexpr (RING) {   // the new plane takes the oldest plane's registers; the queue's phase advances (WAFER_X2_Q1)
                q1[WAFER_X2_Q0(0)][r] = p1new[r];
            } else {
                q1[0][r] = q1[1][r];
                q1[1][r] = q1[2][r];
                q1[2][r] = p1new[r];
            }
        }
        if (XS || act2) {
            auto level2 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                VT res2[RY];
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    res2[r] = zero;
                    if (INTERIOR || (wplane2 && rowwk[r])) {
                        const VT m1 = q1[WAFER_X2_Q1(1)][r];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q1[WAFER_X2_Q1(0)][r][v]; zz[1] = w; zz[2] = (C)q1[WAFER_X2_Q1(2)][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)nbl[1][r];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)nbr[1][r];
                            ys[0] = (r >= 1) ? (C)q1[WAFER_X2_Q1(1)][r >= 1 ? r - 1 : 0][v] : (C)nbu[1][v];
                            ys[2] = (r + 1 < RY) ? (C)q1[WAFER_X2_Q1(1)][r + 1 < RY ? r + 1 : RY - 1][v] : (C)nbd[1][v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = as_stored(update_with(w, (C)caq[r][v], (C)cbq[r][v], S));   // (the sums below see what the array will hold)
                            res2[r][v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                }
                // the sums of this plane: Y1 (still in its queue) against itself and the stored states, Z against the stored
                // states.  Cells outside the work area hold exact zeros at both levels, so nothing is masked here.
                if (!XS || act2)
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double yv = q1[WAFER_X2_Q1(1)][r][v], zv = res2[r][v];
                        acc_y = __builtin_fma(yv, yv, acc_y);
#pragma unroll
                        for (int j = 0; j < NL; ++j) {
                            acc_yl[j] = __builtin_fma(lq[j][r][v], yv, acc_yl[j]);
                            acc_zl[j] = __builtin_fma(lq[j][r][v], zv, acc_zl[j]);
                        }
                    }
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (INTERIOR || rowwk[r]) {
#if WAFER_DIAG & 4   // timing experiment: nothing is stored (the compiler cannot know)
                        if (a.dt > -1.0) continue;
#endif
                        ST *dst = (out + (long long)(XS && zp2 < zs ? zs : zp2) * g.plane + rowoff[r]) + xlu;
                        SVT st2;   // (a storage-type number already: as_stored)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) st2[v] = (ST)res2[r][v];
                        if (INTERIOR || xi + VEC <= g.nx) {
                            wafer_store_result(reinterpret_cast<SVT *>(dst), st2);   // (streamed: wafer_stencil_fused3.hip.h, gstore)
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (xi + v < g.nx) dst[v] = st2[v];
                        }
                    }
                }
            };
            if constexpr (XS) level2(std::true_type{});   // (whole tiles, work planes only: the launcher)
            else if (all_rows && wplane2) level2(std::true_type{});
            else level2(std::false_type{});
        }
        issue_group(3);
        // HOLD: the plane transformed last iteration (z + 1) takes the queue slot just read (same lanes: program order suffices)
        if constexpr (HOLD) {
            QT *qd = qslot(z + 1);
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int r = 0; r < RY; ++r) *reinterpret_cast<QVT *>(qd + j * QS + qoff[r]) = narrow_q(hold_l[j][r]);
        }
        __syncthreads();
        // ---- 5. rotate the pipelines; the plane requested at the top of the iteration is transformed here -- BEHIND the
        //         barrier: the wait for those loads then overlaps the wait for the other waves.  (With the transform down to a
        //         handful of fused multiply-adds the compiler hoisted it above the barrier, and every wave waited for its loads
        //         first and for the slowest wave second: 0.507 against 0.485 ms/step at k = 1.  The pins keep it here.)
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            asm volatile("" : "+v"(pre[r]));
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                asm volatile("" : "+v"(pre_l[j][r]));
                asm volatile("" : "+v"(pre_m[j][r]));
            }
        }
        asm volatile("" : "+v"(xpre));
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            asm volatile("" : "+v"(xpre_l[j]));
            asm volatile("" : "+v"(xpre_m[j]));
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if constexpr (!RING) {
                q0[0][r] = q0[1][r];
                q0[1][r] = q0[2][r];
            }
            VT l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                l[j] = widen(pre_l[j][r]);
                mm[j] = widen(pre_m[j][r]);
                // plane z + 2: read at iteration z + 3, as plane (z + 3) - 1 -- its slot (z + 2) % 3 was last read at iteration z
                if constexpr (HOLD) hold_l[j][r] = l[j];
                else *reinterpret_cast<QVT *>(qslot(z + 2) + j * QS + qoff[r]) = pre_l[j][r];   // (as it arrived: the storage type IS the queue's)
            }
            q0[RING ? WAFER_X2_Q0(0) : 2][r] = xform_vec(widen(pre[r]), l, mm);
            vcur[r] = widen(pre_v[r]);
            caq[r] = canew[r];
            cbq[r] = cbnew[r];
        }
        if constexpr (!RING) {
            xq0[0] = xq0[1];
            xq0[1] = xq0[2];
        }
        if (x_row) {
            VT l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { l[j] = widen(xpre_l[j]); mm[j] = widen(xpre_m[j]); }
            xq0[RING ? WAFER_X2_Q0(0) : 2] = xform_vec(widen(xpre), l, mm);
        } else {
            double l[NL], mm[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) { l[j] = (double)xpre_l[j][0]; mm[j] = (double)xpre_m[j][0]; }
            // (a halo-column lane keeps component 0 only: the slot's other component is never read)
            xq0[RING ? WAFER_X2_Q0(0) : 2][0] = wafer_x2_xform<NL>(kf, (double)xpre[0], l, mm);
        }
        xv = widen(xpre_v);
