// Host-side frame IO of libd2r.so (pngio.cpp): PNG encode / decode and the worker pool the render-and-score pass
// hands its streamed frames to.
#pragma once
#include <stdint.h>

#include <functional>
#include <memory>
#include <string>
#include <vector>

int d2r_png_encode(const uint8_t *rgb, uint32_t w, uint32_t h, int level, std::vector<uint8_t> &out, std::string &err);
int d2r_png_write_file(const uint8_t *rgb, uint32_t w, uint32_t h, int level, const std::string &path, std::string &err);
int d2r_png_decode(const uint8_t *bytes, size_t n, uint32_t want_w, uint32_t want_h, uint8_t *rgb_out, uint32_t *w_out,
                   uint32_t *h_out, std::string &err);
int d2r_png_read_file(const std::string &path, uint32_t want_w, uint32_t want_h, uint8_t *rgb_out, uint32_t *w_out,
                      uint32_t *h_out, std::string &err);
// Frames that equal a known background frame in most scanlines (the frames of a render-and-score pass): the background's scanlines are
// entropy-coded once (d2r_png_base_build), a frame re-codes only the scanlines that differ.  Same pixels as d2r_png_encode.
struct D2rPngBase;
std::shared_ptr<const D2rPngBase> d2r_png_base_build(const uint8_t *bg_rgb, uint32_t w, uint32_t h);
int d2r_png_encode_delta(const D2rPngBase &base, const uint8_t *rgb, std::vector<uint8_t> &out, std::string &err);
int d2r_png_write_file_delta(const D2rPngBase &base, const uint8_t *rgb, const std::string &path, std::string &err);
std::string d2r_png_name(const std::string &dir, uint32_t index);      // <dir>/cb_rgb_%04d.png
int d2r_default_io_threads();

// Worker threads running jobs `int job(std::string &err)`; jobs carry a group id (a staging buffer) so that a buffer
// can be waited for on its own.  The first failing job's code and message are kept for take_error().
#define D2R_POOL_GROUPS 4
class D2rJobPool {
public:
    explicit D2rJobPool(int n_threads);
    ~D2rJobPool();
    D2rJobPool(const D2rJobPool &) = delete;
    D2rJobPool &operator=(const D2rJobPool &) = delete;
    void submit(int group, std::function<int(std::string &)> job);
    void wait(int group);                 // group < 0: every group
    int take_error(std::string &err);     // 0 when no job failed since the last call
    int size() const { return n; }

private:
    struct Impl;
    Impl *impl;
    int n;
};
