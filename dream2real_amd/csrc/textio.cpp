// textio.cpp — the text files the caller of the path persists: goal_pose.txt, pose_batch.txt, pose_scores.txt
// (reference dream2real.py:356-358, np.savetxt with its defaults: fmt '%.18e', delimiter ' ', newline '\n'; read back by
// use_cache_goal_pose :335-341 and use_cache_renders clip_scoring.py:91).  pose_batch.txt of the reference's own grid is
// 70 000 x 16 numbers: np.savetxt formats them one row at a time in Python (over a second); here rows are formatted in
// blocks on the worker pool and written in order.  Byte-identical to np.savetxt: both print the correctly rounded
// 19-significant-digit decimal of the double.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <memory>
#include <string>
#include <vector>

#include "d2r_internal.h"
#include "pngio.h"

static size_t format_e18(char *dst, double v)
{
    if (isnan(v)) {                    // Python prints 'nan' whatever the sign bit; printf may print '-nan'
        memcpy(dst, "nan", 3);
        return 3;
    }
    // std::to_chars(scientific, 18) is specified to produce what printf("%.18e") produces in the C locale (the
    // correctly rounded decimal, exponent of at least two digits) and does it with Ryu-printf instead of glibc's bignums
    auto r = std::to_chars(dst, dst + 40, v, std::chars_format::scientific, 18);
    if (r.ec == std::errc()) return (size_t)(r.ptr - dst);
    return (size_t)snprintf(dst, 40, "%.18e", v);
}

extern "C" int d2r_savetxt(const char *path, const double *data, uint64_t rows, uint64_t cols, int threads)
{
    if (!path || (!data && rows * cols)) return d2r_fail(nullptr, D2R_ERR_INVALID, "null argument");
    if (cols == 0 && rows) return d2r_fail(nullptr, D2R_ERR_INVALID, "d2r_savetxt needs at least one column");
    FILE *f = fopen(path, "wb");
    if (!f) return d2r_fail(nullptr, D2R_ERR_INVALID, std::string("cannot open ") + path + " for writing");
    const uint64_t block = std::max<uint64_t>(1, 65536 / std::max<uint64_t>(1, cols));
    const uint64_t nblocks = (rows + block - 1) / block;
    std::vector<std::string> out(nblocks);
    auto fmt_block = [&](uint64_t b) {
        const uint64_t r0 = b * block, r1 = std::min(rows, r0 + block);
        std::string &s = out[b];
        s.resize((r1 - r0) * cols * 26);
        char *p = &s[0];
        for (uint64_t r = r0; r < r1; r++) {
            for (uint64_t c = 0; c < cols; c++) {
                if (c) *p++ = ' ';
                char tmp[48];
                const size_t n = format_e18(tmp, data[r * cols + c]);
                memcpy(p, tmp, n);
                p += n;
            }
            *p++ = '\n';
        }
        s.resize((size_t)(p - &s[0]));
    };
    if (nblocks <= 1 || threads == 1) {
        for (uint64_t b = 0; b < nblocks; b++) fmt_block(b);
    } else {
        D2rJobPool pool((int)std::min<uint64_t>(nblocks, (uint64_t)(threads > 0 ? threads : d2r_default_io_threads())));
        for (uint64_t b = 0; b < nblocks; b++)
            pool.submit(0, [&fmt_block, b](std::string &) { fmt_block(b); return 0; });
        pool.wait(-1);
    }
    bool ok = true;
    for (auto &s : out) ok = ok && fwrite(s.data(), 1, s.size(), f) == s.size();
    if (fclose(f) != 0 || !ok) return d2r_fail(nullptr, D2R_ERR_INVALID, std::string("short write to ") + path);
    return D2R_OK;
}
