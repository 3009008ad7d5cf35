"""TEST-ONLY level engine: the numpy oracle dressed as ``ptwt_amd._engine.ENGINE``.

The product has no CPU path.  To exercise its *host logic* (axes handling, batch folding, level loop, trim
arithmetic, containers, error types) without a GPU, the CPU tests monkeypatch ``_engine.ENGINE`` with this
object.  It lives under tests/ and is never imported by the package.
"""
import numpy as np
import torch

from oracle import fwt_oracle as O

_MODES = {v: k for k, v in {"zero": 0, "constant": 1, "reflect": 2, "periodic": 3, "symmetric": 4}.items()}


class OracleLevelEngine:
    def analysis(self, x, dec_lo, dec_hi, mode_id):
        ndim = x.dim() - 1
        bank = (np.asarray(dec_lo), np.asarray(dec_hi), None, None)
        bands = O._dwtn(x.detach().numpy(), bank, _MODES[mode_id], list(range(1, ndim + 1)))
        keys = [format(s, f"0{ndim}b").replace("0", "a").replace("1", "d") for s in range(1 << ndim)]
        return torch.from_numpy(np.stack([bands[k] for k in keys], axis=1))

    def analysis_tail(self, x, dec_lo, dec_hi, mode_id, nlevels):
        """Stand-in for the fused deep 1-D levels: same return contract as HipLevelEngine.analysis_tail — every buffer but the last
        holds its detail row only ([B, 1, M]), the last one is laid out like an analysis result ([B, 2, M])."""
        if x.dim() != 2 or x.shape[1] > 64 or nlevels < 2:
            return None
        bufs, cur = [], x
        for _ in range(nlevels):
            buf = self.analysis(cur, dec_lo, dec_hi, mode_id)
            bufs.append(buf)
            cur = buf[:, 0].clone()
        return [b[:, 1:].contiguous() for b in bufs[:-1]] + [bufs[-1]]

    def analysis_pair(self, x, dec_lo, dec_hi, mode_id):
        """Stand-in for the two-levels-per-launch call: same return contract as HipLevelEngine.analysis_pair — the first
        buffer holds the three detail bands only."""
        if x.dim() != 3 or min(x.shape[1:]) < 16:
            return None
        buf1 = self.analysis(x, dec_lo, dec_hi, mode_id)
        buf2 = self.analysis(buf1[:, 0], dec_lo, dec_hi, mode_id)
        return buf1[:, 1:].contiguous(), buf2

    def analysis_pyramid(self, x, dec_lo, dec_hi, mode_id, nlevels):
        """Stand-in for the several-levels-per-launch call: same return contract as HipLevelEngine.analysis_pyramid — every buffer but the
        last holds the three detail bands only.  Like the library it has two routes: planes of at most
        48 x 48 samples get up to eight levels in any mode (the small-plane kernel); bigger ones up to three, rows of a multiple of
        four samples only, every mode but periodic, filters up to 8 taps (the streaming kernel), and may fuse fewer levels than asked."""
        if x.dim() != 3 or x.dtype != torch.float32:
            return None
        small = x.shape[1] * x.shape[2] <= 48 * 48 and len(dec_lo) <= 20
        if not small and (x.shape[2] % 4 or mode_id == 3 or len(dec_lo) > 8):
            return None
        bufs, cur = [], x
        for _ in range(min(nlevels, 8 if small else 3)):
            if not small and min(cur.shape[1:]) < 2 * len(dec_lo):
                break
            buf = self.analysis(cur, dec_lo, dec_hi, mode_id)
            bufs.append(buf)
            cur = buf[:, 0].clone()
        return ([b[:, 1:].contiguous() for b in bufs[:-1]] + [bufs[-1]]) or None

    def pyramid_levels(self, x, flen, mode_id, nlevels):
        """Same contract as HipLevelEngine.pyramid_levels (how many levels analysis_pyramid would fuse; nothing launched): asks the
        stand-in itself on a detached copy."""
        if x.dim() != 3 or x.dtype != torch.float32:
            return 0
        got = self.analysis_pyramid(x.detach(), [0.0] * flen, [0.0] * flen, mode_id, nlevels)
        return 0 if got is None else len(got)

    def synthesis_pyramid_plan(self, approx, levels, flen, out_extent):
        """Same contract as HipLevelEngine.synthesis_pyramid_plan: (plan, descriptors, references, route) — route 1 = the
        whole-reconstruction launch of a small plane (what the stand-in below takes), 0 = no multi-level launch."""
        ok = approx.dim() == 3 and approx.dtype == torch.float32 and out_extent[0] * out_extent[1] <= 48 * 48
        return (None, None, None, 1 if ok and tuple(approx.shape) == tuple(levels[0][0].shape) else 0)

    def synthesis_pyramid(self, approx, levels, rec_lo, rec_hi, out_extent, plan=None):
        """Stand-in for the whole-reconstruction-in-one-launch call (same contract as HipLevelEngine.synthesis_pyramid): planes of at
        most 48 x 48 output samples, the running approximation cropped to the next level's band extents."""
        if approx.dim() != 3 or approx.dtype != torch.float32 or out_extent[0] * out_extent[1] > 48 * 48:
            return None
        if tuple(approx.shape) != tuple(levels[0][0].shape):
            return None
        flen = len(rec_lo)
        cur = approx
        for i, det in enumerate(levels):
            nxt = levels[i + 1][0].shape[1:] if i + 1 < len(levels) else out_extent
            full = [2 * m - flen + 2 for m in cur.shape[1:]]
            if any(n > f or n < 1 for n, f in zip(nxt, full)):
                return None
            cur = self.synthesis(cur, det, rec_lo, rec_hi, full)[:, : nxt[0], : nxt[1]]
        return cur.contiguous()

    def synthesis_tail(self, approx, details, rec_lo, rec_hi, out_lens):
        """Stand-in for the fused coarse 1-D synthesis levels (same contract as HipLevelEngine.synthesis_tail)."""
        if approx.dim() != 2 or max(out_lens) > 64 or len(details) < 2:
            return None
        cur = approx
        for d, n in zip(details, out_lens):
            cur = self.synthesis(cur, [d], rec_lo, rec_hi, [n])
        return cur

    def synthesis_long(self, approx, details, rec_lo, rec_hi, out_lens):
        """Stand-in for the chunked fine 1-D synthesis levels (same contract as HipLevelEngine.synthesis_long): rows of more than 48
        output samples count as "long" here; at most three levels per call, so the come-back-later answer is exercised too."""
        if approx.dim() != 2 or len(details) < 2 or out_lens[-1] <= 48:
            return None, 0
        if len(details) > 3:
            return None, 3
        cur = approx
        for d, n in zip(details, out_lens):
            cur = self.synthesis(cur, [d], rec_lo, rec_hi, [n])
        return cur, len(details)

    def synthesis_pair(self, approx2, details2, details1, rec_lo, rec_hi, out_extent):
        """Stand-in for the two-levels-per-launch synthesis call (same contract as HipLevelEngine.synthesis_pair)."""
        if approx2.dim() != 3 or min(out_extent) < 16:
            return None
        mid = self.synthesis(approx2, details2, rec_lo, rec_hi, list(details1[0].shape[1:]))
        return self.synthesis(mid, details1, rec_lo, rec_hi, out_extent)

    def synthesis(self, approx, details, rec_lo, rec_hi, out_extent):
        ndim = approx.dim() - 1
        flen = len(rec_lo)
        bank = (None, None, np.asarray(rec_lo), np.asarray(rec_hi))
        keys = [format(s, f"0{ndim}b").replace("0", "a").replace("1", "d") for s in range(1 << ndim)]
        bands = {keys[0]: approx.detach().numpy()}
        for k, t in zip(keys[1:], details):
            bands[k] = t.detach().numpy()
        trims = [2 * approx.shape[1 + a] - flen + 2 - out_extent[a] for a in range(ndim)]
        y = O._idwtn(bands, bank, list(range(1, ndim + 1)), trims)
        return torch.from_numpy(np.ascontiguousarray(y))

    # ---- adjoints and tap correlations (test-only, fp64 numpy: the transposes by explicit index arithmetic) ---------------------------
    @staticmethod
    def _pads(n, flen):
        return flen - 2, flen - 2 + n % 2

    def _axis_adjoint(self, g_lo, g_hi, n, lo, hi, mode, axis):
        """Transpose of O.dwt_axis along `axis`: u[e] = sum_k g_b[k] h_b[2 k + 1 - e] on the extended index range, folded back through
        O.ext_index."""
        flen = len(lo)
        g_lo, g_hi = np.moveaxis(g_lo, axis, -1), np.moveaxis(g_hi, axis, -1)
        m = g_lo.shape[-1]
        pl, pr = self._pads(n, flen)
        ext = np.arange(-pl, n + pr)
        u = np.zeros(g_lo.shape[:-1] + (ext.size,), dtype=np.float64)
        for k in range(m):
            for t in range(flen):
                e = 2 * k + 1 - t
                if -pl <= e < n + pr:
                    u[..., e + pl] += g_lo[..., k] * lo[t] + g_hi[..., k] * hi[t]
        src = O.ext_index(ext, n, mode)
        out = np.zeros(g_lo.shape[:-1] + (n,), dtype=np.float64)
        for j, s_ in enumerate(src):
            if s_ >= 0:
                out[..., s_] += u[..., j]
        return np.moveaxis(out, -1, axis)

    def analysis_adjoint(self, g_buf, sig_shape, dec_lo, dec_hi, mode_id):
        nd = g_buf.dim() - 2
        lo, hi = np.asarray(dec_lo, dtype=np.float64), np.asarray(dec_hi, dtype=np.float64)
        bands = {s: g_buf[:, s].detach().numpy().astype(np.float64) for s in range(1 << nd)}
        for a in range(nd):  # undo the axes one by one: pairs that differ in the bit of axis a
            bit = 1 << (nd - 1 - a)
            bands = {s: self._axis_adjoint(t, bands[s | bit], int(sig_shape[a]), lo, hi, _MODES[mode_id], 1 + a) for s, t in bands.items() if not s & bit}
        return torch.from_numpy(np.ascontiguousarray(bands[0])).to(g_buf.dtype)

    def analysis_adjoint_bands(self, g_approx, g_details, sig_shape, dec_lo, dec_hi, mode_id):
        return self.analysis_adjoint(torch.stack([g_approx, *g_details], dim=1), sig_shape, dec_lo, dec_hi, mode_id)

    def synthesis_adjoint(self, g_y, coef_shape, rec_lo, rec_hi):
        """Transpose of synthesis(): g_b[k] = sum_n g_y[n] r_b[n + L - 2 - 2 k] per axis (zeros outside the cropped output)."""
        nd = g_y.dim() - 1
        lo, hi = np.asarray(rec_lo, dtype=np.float64), np.asarray(rec_hi, dtype=np.float64)
        flen = len(lo)
        bands = {0: g_y.detach().numpy().astype(np.float64)}
        for a in reversed(range(nd)):
            bit = 1 << (nd - 1 - a)
            nxt = {}
            for s, t in bands.items():
                t = np.moveaxis(t, 1 + a, -1)
                n, m = t.shape[-1], int(coef_shape[a])
                ga, gd = np.zeros(t.shape[:-1] + (m,)), np.zeros(t.shape[:-1] + (m,))
                for k in range(m):
                    for j in range(flen):
                        i = 2 * k - (flen - 2) + j
                        if 0 <= i < n:
                            ga[..., k] += t[..., i] * lo[j]
                            gd[..., k] += t[..., i] * hi[j]
                nxt[s], nxt[s | bit] = np.moveaxis(ga, -1, 1 + a), np.moveaxis(gd, -1, 1 + a)
            bands = nxt
        return torch.from_numpy(np.stack([bands[s] for s in range(1 << nd)], axis=1)).to(g_y.dtype)

    def tap_correlate(self, a, b, filt_len, c0, sgn, mode_id, out):
        """out[t] += sum_{row, k} a[row, k] b_ext[row, 2 k + c0 + sgn t]"""
        an, bn = a.detach().numpy().astype(np.float64), b.detach().numpy().astype(np.float64)
        n = bn.shape[1]
        for t in range(filt_len):
            idx = O.ext_index(2 * np.arange(an.shape[1]) + c0 + sgn * t, n, _MODES[mode_id])
            vals = np.where(idx >= 0, bn[:, np.clip(idx, 0, n - 1)], 0.0)
            out[t] += float((an * vals).sum())

    def tap_correlate_planes(self, along, a, b, filt_len, c0, sgn, mode_id, out):
        """[batch, rows, columns] operands: along 1 = the reduction of tap_correlate along the columns, 0 = along the rows"""
        if along == 1:
            return self.tap_correlate(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), filt_len, c0, sgn, mode_id, out)
        at, bt = a.transpose(1, 2), b.transpose(1, 2)
        return self.tap_correlate(at.reshape(-1, at.shape[-1]), bt.reshape(-1, bt.shape[-1]), filt_len, c0, sgn, mode_id, out)

    def analysis_outer(self, x, dec_lo, dec_hi, mode_id):
        """[B, N, C] -> (lo, hi) [B, M, C]: one level along the middle axis"""
        xt = x.transpose(1, 2)
        buf = self.analysis(xt.reshape(-1, xt.shape[-1]), dec_lo, dec_hi, mode_id)  # [B * C, 2, M]
        buf = buf.reshape(x.shape[0], x.shape[2], 2, -1)
        return buf[:, :, 0].transpose(1, 2), buf[:, :, 1].transpose(1, 2)

    def synthesis_outer(self, lo, hi, rec_lo, rec_hi, n_out):
        """(lo, hi) [B, M, C] -> [B, n_out, C]: one synthesis level along the middle axis"""
        lt, ht = lo.transpose(1, 2), hi.transpose(1, 2)
        y = self.synthesis(lt.reshape(-1, lt.shape[-1]), [ht.reshape(-1, ht.shape[-1])], rec_lo, rec_hi, [n_out])  # [B * C, n_out]
        return y.reshape(lo.shape[0], lo.shape[2], n_out).transpose(1, 2)

    def tap_correlate_dilated(self, a, b, filt_len, c0, tstep, out):
        """out[t] += sum_{row, k} a[row, k] b[row, (k + c0 + tstep t) mod N]"""
        an, bn = a.detach().numpy().astype(np.float64), b.detach().numpy().astype(np.float64)
        n = bn.shape[1]
        for t in range(filt_len):
            out[t] += float((an * bn[:, (np.arange(n) + c0 + tstep * t) % n]).sum())


def swt_level_fwd(x, lo, hi, dilation, scale):
    """TEST-ONLY stand-in for stationary_transform._level_fwd: buf[:, 0 / 1][n] = s sum_m lo / hi[m] x[(n + D (L/2 - m)) mod N]."""
    xn = x.detach().numpy().astype(np.float64)
    n, flen = xn.shape[1], len(lo)
    out = np.zeros((xn.shape[0], 2, n))
    for m in range(flen):
        sh = xn[:, (np.arange(n) + dilation * (flen // 2 - m)) % n]
        out[:, 0] += scale * lo[m] * sh
        out[:, 1] += scale * hi[m] * sh
    return torch.from_numpy(out).to(x.dtype)


def swt_level_inv(a, d, lo, hi, dilation, scale):
    """TEST-ONLY stand-in for stationary_transform._level_inv: y[n] = s sum_j lo[j] a[(n + D (L/2 - 1 - j)) mod N] + hi[j] d[...]."""
    an, dn = a.detach().numpy().astype(np.float64), d.detach().numpy().astype(np.float64)
    n, flen = an.shape[1], len(lo)
    y = np.zeros_like(an)
    for j in range(flen):
        idx = (np.arange(n) + dilation * (flen // 2 - 1 - j)) % n
        y += scale * (lo[j] * an[:, idx] + hi[j] * dn[:, idx])
    return torch.from_numpy(y).to(a.dtype)
