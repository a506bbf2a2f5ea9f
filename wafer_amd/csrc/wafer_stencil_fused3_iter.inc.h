// One plane iteration of the three-step kernel (wafer_stencil_fused3.hip.h, wafer_step3_body): included as TEXT into the plane
// loop -- three times, with WAFER_F3_PH = 0, 1, 2, where the z-queues are rings (RING) -- because as a lambda the body costs the
// peer instantiation its registers (profiles/NOTES.md).  `it` is the iteration; everything else is the enclosing function's.
// RING: the loop is unrolled by three and logical plane m of a z-queue lives in physical slot (m + phase) % 3; a new plane takes
// the slot of the oldest one and the phase advances, so the 16 vector shifts per iteration (32 v_mov_b64 of ~420 vector
// instructions) disappear.  phi1 / phi2 queues advance in mid-iteration (before the level that reads them): WAFER_F3_Q1.
        const int z = z1 + SD * it;
        const bool more = it + 1 < niter;
#if WAFER_DIAG & 2   // timing experiment: every prefetch asks for the column's first planes again (cache hits)
        const long long zo = (long long)(z1 + (it & 1)) * g.plane;
#else
        const long long zo = (long long)z * g.plane;
#endif
#ifndef WAFER_F3_LATE_WAIT_HOISTED   // (the tail segment of a segmented pass waits once, ahead of its loop: wafer_step3_body)
        if constexpr (SYNC) {
            if (blk.wait_late >= 0 && it == blk.wait_it) poisoned = wafer_f3_wait(syv, blk.wait_late, tid);
        }
#endif
        // ---- 1. prefetch: phi0 two planes ahead, V one plane ahead
        // DIRECT (with RING): what is dead by the time its successor is requested takes the request itself -- V of the main rows
        // (last read by level 1, requested behind it), the extra slot's V, oldest phi0 plane and outer-row copy (last read by the extra
        // slot's level 1 / the staging at the top, requested behind them): no staging registers and no copies for these four.
        // (fp32 storage: what arrives is float and is widened where the queues rotate, behind the barrier -- a request cannot
        //  write the slot itself)
        constexpr bool DIRECT = RING && !WIDE;
        SVT szero;
#pragma unroll
        for (int v = 0; v < VEC; ++v) szero[v] = ST(0);
        SVT pre[RY], pre_v[RY], xpre = szero, xpre_v = szero, orow_pre = szero;
#pragma unroll
        for (int r = 0; r < RY; ++r) pre[r] = pre_v[r] = szero;
        // The seven requests of a wave are NOT issued together: all eight waves leave the barrier at once, and 56 requests of 1 KiB
        // queue at the CU's one address unit (16 cycles each) while no wave can issue arithmetic behind its own -- the in-kernel
        // stamps (tools/f3_stamps.py) showed a sixth of the iteration going there.  Spread over the iteration (the main rows' phi0
        // at the top, their V behind level 1, the extra slot's three behind level 2) they overlap the other waves' arithmetic:
        // 0.249 -> 0.236 ms/step at 512^3, and the later requests hold their registers for a shorter time (244 -> 230 VGPRs).
        // A: phi0 of the main rows, at the top; B: their V, behind level 1 of the main rows; C: the extra slot's three, behind level 1
        // of the extra slot (where the placements were measured: profiles/r04_ab_f3_request_placement.jsonl)
        auto issue_A = [&]() {
#pragma unroll
            for (int r = 0; r < RY; ++r) pre[r] = gload_raw((phi + zo + SD * 2 * g.plane + rowoff[r]) + xlu);
        };
        auto issue_B = [&]() {
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                if constexpr (DIRECT) vcur[r] = gload((pv + zo + SD * g.plane + rowoff[r]) + xlu);
                else pre_v[r] = gload_raw((pv + zo + SD * g.plane + rowoff[r]) + xlu);
            }
        };
        auto issue_C = [&]() {
            if constexpr (DIRECT) {
                xq0[WAFER_F3_Q0(0)] = gload(phi + zo + SD * 2 * g.plane + xslot_off);
                xv = gload(pv + zo + SD * g.plane + xslot_off);
                orow_nxt = gload(phi + zo + SD * 2 * g.plane + orow_slot_off);
            } else {
                xpre = gload_raw(phi + zo + SD * 2 * g.plane + xslot_off);
                xpre_v = gload_raw(pv + zo + SD * g.plane + xslot_off);
                orow_pre = gload_raw(phi + zo + SD * 2 * g.plane + orow_slot_off);
            }
        };
        WAFER_F3_SETPRIO(3);
        issue_A();
        // The extra slot's requests are the SAME three instructions in every wave, the address chosen per lane (a halo row's
        // 16 bytes, or the 16 bytes that start at the lane's halo-column cell: component 0 is the cell; a wave without an outer
        // row asks for its slot's line again).  As two branches with loads of their own -- row waves / column waves -- the
        // compiler let the branches share destination registers, and its wait-count pass, which cannot know that a wave takes
        // one branch for life, then made the row waves wait for every request in flight before they issued theirs: the four row
        // waves of every workgroup sat out the memory latency at the top of each iteration (0.279 against 0.258 ms/step at
        // 512^3; which builds fell into it depended on the register allocator's mood).
        // ---- 2. stage the next phi0 plane into the other buffer
        if (more) {
            T *nt = lds0 + ((z + 1) & 1) * Cfg::TILE0;
#pragma unroll
            for (int r = 0; r < RY; ++r) *reinterpret_cast<VT *>(nt + (yrow[r] - (y0 - 3)) * LP0 + HX0 + xl) = q0[WAFER_F3_Q0(2)][r];
            // (what was requested in place of a cell outside the work area becomes the zero it stands for HERE, where the value
            //  is used, not where the request is waited for: a select on the prefetch registers before the barrier would put
            //  the wait into the iteration that issued the request.  The queues of such a cell are never read otherwise:
            //  xwk / c_wk.)
            if (x_row) {
                *reinterpret_cast<VT *>(nt + (xy - (y0 - 3)) * LP0 + HX0 + xl) = xy_out ? zero : xq0[WAFER_F3_Q0(2)];
                if (has_orow) *reinterpret_cast<VT *>(nt + orow_lds) = oy_out ? zero : orow_nxt;
            } else if (c_ok) nt[c_lds0] = c_xout ? T(0) : xq0[WAFER_F3_Q0(2)][0];
        }
        WAFER_F3_STAMP_AT(0);   // requests issued, next plane staged
        const T *c0 = lds0 + (z & 1) * Cfg::TILE0;
        T *w1 = lds1 + (z & 1) * Cfg::TILE1;
        const T *c1 = lds1 + ((z + 1) & 1) * Cfg::TILE1;
        T *w2 = lds2 + ((z + 1) & 1) * Cfg::TILE2;
        const T *c2 = lds2 + (z & 1) * Cfg::TILE2;
        const bool wplane1 = work_plane(z), wplane2 = work_plane(z - SD);
        const int zp2 = z - SD;
        const bool need2 = zp2 >= zs - 1 && zp2 <= ze;   // phi2 is read on planes zs-1 .. ze only
        VT p1new[RY], p2new[RY], canew[RY], cbnew[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) p1new[r] = p2new[r] = canew[r] = cbnew[r] = zero;
        VT xp1 = zero, xcanew = zero, xcbnew = zero;

        bool all_rows = x0 + TX <= g.nx;
#pragma unroll
        for (int r = 0; r < RY; ++r) all_rows = all_rows && rowwk[r];
        // ---- 2b. the x / y neighbours of the main rows, requested one level AHEAD: level 1's and level 2's here, level 3's
        //          behind level 1's arithmetic.  What level L reads this iteration (phi0 plane z, phi1 plane z-1, phi2 plane z-2)
        //          was written before the last barrier, and the ring slots the levels write are the other ones: reading early
        //          changes no value.  In program order every level's reads sat behind the previous level's ring write (which the
        //          compiler cannot tell apart from them), so each level paid its own LDS round trip with only one other wave on
        //          the SIMD to hide it.  Per level: the cell left of the lane's first and right of its last on each row, the
        //          row above the first row and the row below the last.  (All three levels at the top: 255 VGPRs and scratch.)
        T nbl[3][RY], nbr[3][RY];
        VT nbu[3], nbd[3];
        auto nbload = [&](auto level_tag) {
            constexpr int L = decltype(level_tag)::value;
            const T *cc = L == 0 ? c0 : L == 1 ? c1 : c2;
            constexpr int lp = L == 0 ? LP0 : L == 1 ? LP1 : LP2, hx = L == 0 ? HX0 : L == 1 ? HX1 : HX2;
            const int yb = y0 - 3 + L;
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
        nbload(std::integral_constant<int, 1>{});
        // ---- 3. level 1, main rows.  INTERIOR: the plane and both rows are work cells, the tile's columns too: no tests
        //         inside, the RY x VEC updates form one basic block
        auto level1 = [&](auto interior_tag) {
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int r = 0; r < RY; ++r) {
                VT res = zero;
                if (INTERIOR || (wplane1 && rowwk[r])) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const C w = (C)q0[WAFER_F3_Q0(1)][r][v];
                        C xs[3], ys[3], zz[3];
                        zz[0] = (C)q0[WAFER_F3_Q0(ZLO)][r][v]; zz[1] = w; zz[2] = (C)q0[WAFER_F3_Q0(ZHI)][r][v];
                        xs[1] = ys[1] = w;
                        xs[0] = (v >= 1) ? (C)q0[WAFER_F3_Q0(1)][r][(v + VEC - 1) % VEC] : (C)nbl[0][r];
                        xs[2] = (v + 1 < VEC) ? (C)q0[WAFER_F3_Q0(1)][r][(v + 1) % VEC] : (C)nbr[0][r];
                        ys[0] = (r >= 1) ? (C)q0[WAFER_F3_Q0(1)][r >= 1 ? r - 1 : 0][v] : (C)nbu[0][v];
                        ys[2] = (r + 1 < RY) ? (C)q0[WAFER_F3_Q0(1)][r + 1 < RY ? r + 1 : RY - 1][v] : (C)nbd[0][v];
                        const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                        C ka, kb;
                        const T rs = update_keep(w, (C)vcur[r][v], S, ka, kb);
                        canew[r][v] = (T)ka;
                        cbnew[r][v] = (T)kb;
                        res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                    }
                }
                p1new[r] = res;
                *reinterpret_cast<VT *>(w1 + (yrow[r] - (y0 - 2)) * LP1 + HX1 + xl) = res;
            }
        };
        if (all_rows && wplane1) level1(std::true_type{});
        else level1(std::false_type{});
        WAFER_F3_SETPRIO(2);
        WAFER_F3_STAMP_AT(1);   // level 1, main rows (with the neighbours' LDS round trip)
        issue_B();
        nbload(std::integral_constant<int, 2>{});
        // ---- 3x. level 1, the extra slot
        if (x_row) {
            VT res = zero;
            if (wplane1 && xwk) {
                const int ly = xy - (y0 - 3);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const C w = (C)xq0[WAFER_F3_Q0(1)][v];
                    C xs[3], ys[3], zz[3];
                    zz[0] = (C)xq0[WAFER_F3_Q0(ZLO)][v]; zz[1] = w; zz[2] = (C)xq0[WAFER_F3_Q0(ZHI)][v];
                    xs[1] = ys[1] = w;
                    xs[0] = (v >= 1) ? (C)xq0[WAFER_F3_Q0(1)][(v + VEC - 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v - 1];
                    xs[2] = (v + 1 < VEC) ? (C)xq0[WAFER_F3_Q0(1)][(v + 1) % VEC] : (C)c0[ly * LP0 + HX0 + xl + v + 1];
                    ys[0] = (C)c0[(ly - 1) * LP0 + HX0 + xl + v];
                    ys[2] = (C)c0[(ly + 1) * LP0 + HX0 + xl + v];
                    const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                    C ka, kb;
                    const T rs = update_keep(w, (C)xv[v], S, ka, kb);
                    xcanew[v] = (T)ka;
                    xcbnew[v] = (T)kb;
                    res[v] = (xi + v < g.nx) ? rs : T(0);
                }
            }
            xp1 = res;
            *reinterpret_cast<VT *>(w1 + (xy - (y0 - 2)) * LP1 + HX1 + xl) = res;
        } else if (c_l1) {
            T rs = T(0);
            if (wplane1 && c_wk) {
                const C w = (C)xq0[WAFER_F3_Q0(1)][0];
                C xs[3], ys[3], zz[3];
                zz[0] = (C)xq0[WAFER_F3_Q0(ZLO)][0]; zz[1] = w; zz[2] = (C)xq0[WAFER_F3_Q0(ZHI)][0];
                xs[1] = ys[1] = w;
                xs[0] = (C)c0[c_lds0 - 1]; xs[2] = (C)c0[c_lds0 + 1];
                ys[0] = (C)c0[c_lds0 - LP0]; ys[2] = (C)c0[c_lds0 + LP0];
                const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                C ka, kb;
                rs = update_keep(w, (C)xv[0], S, ka, kb);
                xcanew[0] = (T)ka;
                xcbnew[0] = (T)kb;
            }
            w1[c_lds1] = rs;
            xp1[0] = rs;
        }
        WAFER_F3_SETPRIO(1);
        issue_C();
        WAFER_F3_STAMP_AT(2);   // level 1, the extra slot
        // ---- 4. level 2: phi2 of the plane behind, from the phi1 queues; a, b as level 1 formed them one iteration ago
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if constexpr (RING) {   // the new plane takes the oldest plane's registers; the queue's phase advances (WAFER_F3_Q1)
                q1[WAFER_F3_Q0(0)][r] = p1new[r];
            } else {
                q1[0][r] = q1[1][r];
                q1[1][r] = q1[2][r];
                q1[2][r] = p1new[r];
            }
        }
        if constexpr (RING) {
            xq1[WAFER_F3_Q0(0)] = xp1;
        } else {
            xq1[0] = xq1[1];
            xq1[1] = xq1[2];
            xq1[2] = xp1;
        }
        if (need2) {
            auto level2 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    VT res = zero;
                    if (INTERIOR || (wplane2 && rowwk[r])) {
                        const VT m1 = q1[WAFER_F3_Q1(1)][r];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q1[WAFER_F3_Q1(ZLO)][r][v]; zz[1] = w; zz[2] = (C)q1[WAFER_F3_Q1(ZHI)][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)nbl[1][r];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)nbr[1][r];
                            ys[0] = (r >= 1) ? (C)q1[WAFER_F3_Q1(1)][r >= 1 ? r - 1 : 0][v] : (C)nbu[1][v];
                            ys[2] = (r + 1 < RY) ? (C)q1[WAFER_F3_Q1(1)][r + 1 < RY ? r + 1 : RY - 1][v] : (C)nbd[1][v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update_with(w, (C)caq[1][r][v], (C)cbq[1][r][v], S);
                            res[v] = (INTERIOR || xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    p2new[r] = res;
                    *reinterpret_cast<VT *>(w2 + (yrow[r] - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            };
            if (all_rows && wplane2) level2(std::true_type{});
            else level2(std::false_type{});
            if (x_row) {
                if (x_l2) {
                    VT res = zero;
                    if (wplane2 && xwk) {
                        const int ly = xy - (y0 - 2);
                        const VT m1 = xq1[WAFER_F3_Q1(1)];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)m1[v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)xq1[WAFER_F3_Q1(ZLO)][v]; zz[1] = w; zz[2] = (C)xq1[WAFER_F3_Q1(ZHI)][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)m1[(v + VEC - 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v - 1];
                            xs[2] = (v + 1 < VEC) ? (C)m1[(v + 1) % VEC] : (C)c1[ly * LP1 + HX1 + xl + v + 1];
                            ys[0] = (C)c1[(ly - 1) * LP1 + HX1 + xl + v];
                            ys[2] = (C)c1[(ly + 1) * LP1 + HX1 + xl + v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update_with(w, (C)xca[v], (C)xcb[v], S);
                            res[v] = (xi + v < g.nx) ? rs : T(0);
                        }
                    }
                    *reinterpret_cast<VT *>(w2 + (xy - (y0 - 1)) * LP2 + HX2 + xl) = res;
                }
            } else if (c_l2) {
                T rs = T(0);
                if (wplane2 && c_wk) {
                    const C w = (C)xq1[WAFER_F3_Q1(1)][0];
                    C xs[3], ys[3], zz[3];
                    zz[0] = (C)xq1[WAFER_F3_Q1(ZLO)][0]; zz[1] = w; zz[2] = (C)xq1[WAFER_F3_Q1(ZHI)][0];
                    xs[1] = ys[1] = w;
                    xs[0] = (C)c1[c_lds1 - 1]; xs[2] = (C)c1[c_lds1 + 1];
                    ys[0] = (C)c1[c_lds1 - LP1]; ys[2] = (C)c1[c_lds1 + LP1];
                    const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                    rs = update_with(w, (C)xca[0], (C)xcb[0], S);
                }
                w2[c_lds2] = rs;
            }
        }
        WAFER_F3_SETPRIO(0);
        WAFER_F3_STAMP_AT(3);   // level 2 (main rows and the extra slot)
        // ---- 5. level 3: phi3 two planes behind from the phi2 queue, a, b as formed two iterations ago; stored
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if constexpr (RING) {
                q2[WAFER_F3_Q0(0)][r] = p2new[r];
            } else {
                q2[0][r] = q2[1][r];
                q2[1][r] = q2[2][r];
                q2[2][r] = p2new[r];
            }
        }
        const int zo3 = z - 2 * SD;
        const bool last_wt = SYNC && blk.bump >= 0 && (DOWN ? zo3 < zs + blk.wt : zo3 >= ze - blk.wt);      // the last wt planes of the march
        const bool first_wt = PEER && bump_early >= 0 && (DOWN ? zo3 >= ze - blk.wt : zo3 < zs + blk.wt);   // the first wt planes (whole-column peer passes)
        // mode 2: the planes the exchange kernel reads while this kernel is still running go to memory at once; peer mode: nobody
        // reads them before the kernel ends, what travels is the copy into the neighbour's ghost planes
        const bool wthrough = last_wt && !PEER;
        // the neighbour's buffer, shifted so that this slab's plane index addresses the ghost plane it fills (nullptr: no peer stores).
        // Stored from the registers, inside the loop's store path: a copy from memory after the fact (the planes read back at agent
        // scope, two more barriers per boundary) measured 0.320 against 0.280 ms/step at the bench slab.
        ST *peer_dst = nullptr;
        if constexpr (PEER) peer_dst = first_wt ? peer_first : last_wt ? peer_last : nullptr;
        if (XS || (zo3 >= zs && zo3 < ze)) {
            auto level3 = [&](auto interior_tag) {
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                VT res3[RY];
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    res3[r] = zero;
                    if (INTERIOR || rowwk[r]) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const C w = (C)q2[WAFER_F3_Q1(1)][r][v];
                            C xs[3], ys[3], zz[3];
                            zz[0] = (C)q2[WAFER_F3_Q1(ZLO)][r][v]; zz[1] = w; zz[2] = (C)q2[WAFER_F3_Q1(ZHI)][r][v];
                            xs[1] = ys[1] = w;
                            xs[0] = (v >= 1) ? (C)q2[WAFER_F3_Q1(1)][r][(v + VEC - 1) % VEC] : (C)nbl[2][r];
                            xs[2] = (v + 1 < VEC) ? (C)q2[WAFER_F3_Q1(1)][r][(v + 1) % VEC] : (C)nbr[2][r];
                            ys[0] = (r >= 1) ? (C)q2[WAFER_F3_Q1(1)][r - 1 < 0 ? 0 : r - 1][v] : (C)nbu[2][v];
                            ys[2] = (r + 1 < RY) ? (C)q2[WAFER_F3_Q1(1)][r + 1 < RY ? r + 1 : RY - 1][v] : (C)nbd[2][v];
                            const C S = wafer_stencil_sum<C, 1>(xs, ys, zz, w);
                            const T rs = update_with(w, (C)caq[0][r][v], (C)cbq[0][r][v], S);
                            res3[r][v] = (SYNC && poisoned) ? (T)__builtin_nanf("") : rs;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RY; ++r) {
                    if (INTERIOR || rowwk[r]) {
#if WAFER_DIAG & 4   // timing experiment: nothing is stored; level 3's results stay live (a test on a kernel argument let the
                     // compiler sink level 3's arithmetic behind the test -- the round-5 "no stores" figure had lost a fifth of the
                     // vector work with the stores)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) asm volatile("" ::"v"(res3[r][v]));
                        continue;
#endif
                        const int zst = !XS ? zo3 : DOWN ? (zo3 < ze - 1 ? zo3 : ze - 1) : (zo3 > zs ? zo3 : zs);
                        ST *dst = (out + (long long)zst * g.plane + rowoff[r]) + xlu;
                        SVT st3;   // (the value is a storage-type number already: as_stored)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) st3[v] = (ST)res3[r][v];
                        if constexpr (PEER) {
                            // (XS: level 3 also runs while the pipeline fills; what it produces then goes nowhere near a neighbour)
                            if (peer_dst && (!XS || (zo3 >= zs && zo3 < ze))) {   // into the neighbour's ghost planes (wave-uniform), system scope, written through
                                ST *pd = (peer_dst + (long long)zo3 * g.plane + rowoff[r]) + xlu;
#pragma unroll
                                for (int v = 0; v < VEC; ++v)
                                    if (INTERIOR || xi + v < g.nx) __hip_atomic_store(pd + v, st3[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                        }
                        if constexpr (XS) {
                            // (overlap mode 2's tail: the planes the exchange reads while this kernel runs are written through)
                            if (!(SYNC && !PEER) || !wthrough) {
                                gstore(dst, st3);
                                continue;
                            }
                        }
                        if (wthrough) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (INTERIOR || xi + v < g.nx) __hip_atomic_store(dst + v, st3[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else if (INTERIOR || xi + VEC <= g.nx) {
                            gstore(dst, st3);
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (xi + v < g.nx) dst[v] = st3[v];
                        }
                    }
                }
            };
            if constexpr (XS) level3(std::true_type{});   // (grids of whole tiles only: the launcher)
            else if (all_rows) level3(std::true_type{});
            else level3(std::false_type{});
        }
        WAFER_F3_STAMP_AT(4);   // level 3 and its stores
        // whole-column peer passes: the first wt planes are out after iteration wt + 3 -- acknowledged here, counted behind the barrier
        const bool early_done = PEER && bump_early >= 0 && it == blk.wt + 3;
        if constexpr (PEER) {
            if (early_done) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if WAFER_DIAG & 8   // timing experiment: no workgroup barrier in the plane loop (LDS contents race)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
        __syncthreads();
#endif
        if constexpr (PEER) {
            if (early_done && tid == 0) {
                if (flag_early) __hip_atomic_fetch_add(flag_early, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        WAFER_F3_STAMP_AT(5);   // the barrier
        // ---- 6. rotate the phi0 / V / a, b pipelines.  The prefetched values are pinned HERE, behind the barrier: left to itself the
        //         compiler sometimes consumes a prefetch where it was issued (the halo-column waves then wait out the whole memory
        //         latency at the top of every iteration) or ahead of the barrier (every wave waits for its loads first and for the
        //         slowest wave second) -- which of the two 8 % apart "states" a build landed in used to depend on unrelated edits.
        auto pin = [](auto &x) {
            if constexpr (VEC == 1) {
                auto t = x[0];
                asm volatile("" : "+v"(t));
                x[0] = t;
            } else asm volatile("" : "+v"(x));
        };
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            pin(pre[r]);
            if constexpr (DIRECT) pin(vcur[r]);
            else pin(pre_v[r]);
        }
        if constexpr (DIRECT) {
            pin(xq0[WAFER_F3_Q0(0)]);
            pin(xv);
            pin(orow_nxt);
        } else {
            pin(xpre);
            pin(xpre_v);
            pin(orow_pre);
        }
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            if constexpr (RING) {
                q0[WAFER_F3_Q0(0)][r] = widen(pre[r]);
            } else {
                q0[0][r] = q0[1][r];
                q0[1][r] = q0[2][r];
                q0[2][r] = widen(pre[r]);
            }
            if constexpr (!DIRECT) vcur[r] = widen(pre_v[r]);
            caq[0][r] = caq[1][r];
            cbq[0][r] = cbq[1][r];
            caq[1][r] = canew[r];
            cbq[1][r] = cbnew[r];
        }
        if constexpr (DIRECT) {
            // (requested into the slot itself)
        } else if constexpr (RING) {
            xq0[WAFER_F3_Q0(0)] = widen(xpre);
        } else {
            xq0[0] = xq0[1];
            xq0[1] = xq0[2];
            xq0[2] = widen(xpre);
        }
        if constexpr (!DIRECT) xv = widen(xpre_v);
        xca = xcanew;
        xcb = xcbnew;
        if constexpr (!DIRECT) orow_nxt = widen(orow_pre);
#if WAFER_DIAG & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (what is left of the requests' latency, made visible)
#endif
        WAFER_F3_STAMP_AT(6);   // the wait for the prefetched planes, the queue rotation
