// Frame output of the reference's loop: inference/nerfs/base_nerf_infer.py:97-101 writes every rendered frame as
// `<tmp_imgs_dir>/<idx:05d>.png` (cv2.imwrite of the uint8 picture), synchronously, between two frames.  Here: a pool of native worker
// threads (no interpreter lock anywhere on the path) that deflate and write the frames while the GPU renders the next ones.  Pure host
// code; zlib does the DEFLATE stream and the CRCs.  PNG: 8-bit RGB, colour type 2, filter type 0 on every row -- decodable by any reader.
//
// Round 3: the Python thread pool this replaces (zlib.compress on 32 threads) delivered 0.75-0.84 of the render rate: the workers need the
// interpreter lock between their zlib calls, and the render thread holds it for most of a 0.7-1.3 ms frame.
#include "common.hpp"
#include "geneface_hip.h"

#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <stdio.h>
#include <string.h>

namespace {

using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

inline void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void chunk(std::vector<uint8_t>& out, const char tag[4], const uint8_t* data, size_t n) {
    put32(out, (uint32_t)n);
    const size_t at = out.size();
    out.insert(out.end(), tag, tag + 4);
    if (n) out.insert(out.end(), data, data + n);
    put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

// rgb [H, W, 3] -> a complete PNG file in `out`.  The scanlines go to deflate() row by row (filter byte, then the row): no staging copy.
int encode(const uint8_t* rgb, uint32_t H, uint32_t W, int level, int strategy, std::vector<uint8_t>& out, std::vector<uint8_t>& z) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, strategy) != Z_OK) return -1;
    const size_t row = (size_t)3 * W, raw = (size_t)H * (row + 1);
    z.resize(deflateBound(&zs, (uLong)raw));
    zs.next_out = z.data();
    zs.avail_out = (uInt)z.size();
    const uint8_t zero = 0;
    for (uint32_t y = 0; y < H; y++) {
        zs.next_in = const_cast<Bytef*>(&zero); zs.avail_in = 1;
        if (deflate(&zs, Z_NO_FLUSH) != Z_OK) { deflateEnd(&zs); return -1; }
        zs.next_in = const_cast<Bytef*>(rgb + (size_t)y * row); zs.avail_in = (uInt)row;
        if (deflate(&zs, y + 1 == H ? Z_FINISH : Z_NO_FLUSH) < 0) { deflateEnd(&zs); return -1; }
    }
    if (H == 0) (void)deflate(&zs, Z_FINISH);
    const size_t zn = zs.total_out;
    deflateEnd(&zs);
    out.clear();
    out.reserve(zn + 64);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, W); put32(ihdr, H);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};     // bit depth 8, colour type 2 (RGB), deflate, adaptive filtering (type 0 rows), no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zn);
    chunk(out, "IEND", nullptr, 0);
    return 0;
}

struct Job { uint32_t idx; std::unique_ptr<uint8_t[]> rgb; };

struct Writer {
    std::string dir;
    uint32_t H, W;
    int level, strategy;
    size_t max_pending;
    std::vector<std::thread> threads;
    std::deque<Job> queue;
    std::mutex mu;
    std::condition_variable cv_job, cv_room;
    bool closing = false;
    std::atomic<int> failed{0};   // set AFTER err is complete (release), read with acquire: a reader that sees 1 sees the whole message
    char err[256] = "";
    std::mutex err_mu;

    void fail(const char* what, const char* path, uint32_t idx) {
        std::lock_guard<std::mutex> g(err_mu);
        if (failed.load(std::memory_order_relaxed)) return;   // keep the first error
        if (path) snprintf(err, sizeof(err), "png_writer: %s %s", what, path);
        else snprintf(err, sizeof(err), "png_writer: %s %u", what, idx);
        failed.store(1, std::memory_order_release);
    }
    // per-stage seconds summed over the workers (and the submit side's copy), for the bench line
    std::mutex st_mu;
    double t_copy = 0, t_deflate = 0, t_write = 0, t_wait_room = 0;
    uint64_t n_done = 0, bytes_out = 0;

    void run() {
        std::vector<uint8_t> out, z;
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return closing || !queue.empty(); });
                if (queue.empty()) return;
                job = std::move(queue.front());
                queue.pop_front();
            }
            cv_room.notify_one();
            const auto t0 = clk::now();
            const int rc = encode(job.rgb.get(), H, W, level, strategy, out, z);
            const auto t1 = clk::now();
            bool ok = rc == 0;
            if (ok) {
                char name[32];
                snprintf(name, sizeof(name), "/%05u.png", job.idx);
                const std::string path = dir + name;
                FILE* fh = fopen(path.c_str(), "wb");
                ok = fh && fwrite(out.data(), 1, out.size(), fh) == out.size();
                if (fh) ok = (fclose(fh) == 0) && ok;
                if (!ok) fail("cannot write", path.c_str(), job.idx);
            } else {
                fail("deflate failed on frame", nullptr, job.idx);
            }
            const auto t2 = clk::now();
            std::lock_guard<std::mutex> g(st_mu);
            t_deflate += secs(t0, t1); t_write += secs(t1, t2); n_done++; bytes_out += out.size();
        }
    }
};

}  // namespace

GF_EXPORT void* gf_png_writer_create(const char* dir, uint32_t H, uint32_t W, uint32_t workers, int level, int strategy, uint32_t max_pending) {
    if (!dir || H == 0 || W == 0 || workers == 0 || workers > 256 || level < 0 || level > 9 || strategy < 0 || strategy > 4) {
        gf_set_error(GF_ERR_INVALID, "png_writer: bad argument");
        return nullptr;
    }
    Writer* w = new Writer();
    w->dir = dir; w->H = H; w->W = W; w->level = level; w->strategy = strategy;
    w->max_pending = max_pending ? max_pending : 4u * workers;
    for (uint32_t i = 0; i < workers; i++) w->threads.emplace_back([w] { w->run(); });
    return w;
}

// Copies the frame (callers hand in a pinned buffer the render loop is about to reuse) and queues it; blocks while max_pending frames wait.
GF_EXPORT int gf_png_writer_submit(void* handle, uint32_t idx, const uint8_t* rgb_host) {
    Writer* w = reinterpret_cast<Writer*>(handle);
    if (!w || !rgb_host) return gf_set_error(GF_ERR_INVALID, "png_writer: null pointer");
    const size_t n = (size_t)w->H * w->W * 3;
    const auto t0 = clk::now();
    Job job{idx, std::unique_ptr<uint8_t[]>(new uint8_t[n])};
    memcpy(job.rgb.get(), rgb_host, n);
    const auto t1 = clk::now();
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv_room.wait(lk, [&] { return w->queue.size() < w->max_pending; });
        w->queue.push_back(std::move(job));
    }
    w->cv_job.notify_one();
    const auto t2 = clk::now();
    {
        std::lock_guard<std::mutex> g(w->st_mu);
        w->t_copy += secs(t0, t1); w->t_wait_room += secs(t1, t2);
    }
    return w->failed.load(std::memory_order_acquire) ? gf_set_error(GF_ERR_INVALID, "%s", w->err) : GF_OK;
}

// Waits for every queued frame, joins the workers, frees the writer.  stats_out_host (or NULL) [6]: seconds spent copying frames in (submit
// side), waiting for queue room (submit side), deflating (summed over workers), writing files (summed over workers), frames written, bytes written.
GF_EXPORT int gf_png_writer_close(void* handle, double* stats_out_host) {
    Writer* w = reinterpret_cast<Writer*>(handle);
    if (!w) return gf_set_error(GF_ERR_INVALID, "png_writer: null handle");
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->closing = true;
    }
    w->cv_job.notify_all();
    for (auto& t : w->threads) t.join();
    if (stats_out_host) {
        stats_out_host[0] = w->t_copy; stats_out_host[1] = w->t_wait_room; stats_out_host[2] = w->t_deflate; stats_out_host[3] = w->t_write;
        stats_out_host[4] = (double)w->n_done; stats_out_host[5] = (double)w->bytes_out;
    }
    const int failed = w->failed.load(std::memory_order_acquire);
    int rc = GF_OK;
    if (failed) rc = gf_set_error(GF_ERR_INVALID, "%s", w->err);
    delete w;
    return rc;
}

// One picture, synchronously: rgb_host [H, W, 3] -> PNG bytes in out_host (capacity cap_bytes); *n_bytes_host = size written.
GF_EXPORT int gf_png_encode_rgb8(const uint8_t* rgb_host, uint32_t H, uint32_t W, int level, int strategy, uint8_t* out_host, uint64_t cap_bytes,
                                 uint64_t* n_bytes_host) {
    if (!rgb_host || !out_host || !n_bytes_host || H == 0 || W == 0) return gf_set_error(GF_ERR_INVALID, "png_encode: bad argument");
    std::vector<uint8_t> out, z;
    if (encode(rgb_host, H, W, level, strategy, out, z)) return gf_set_error(GF_ERR_INVALID, "png_encode: deflate failed");
    if (out.size() > cap_bytes) return gf_set_error(GF_ERR_INVALID, "png_encode: output buffer too small (%zu > %llu)", out.size(), (unsigned long long)cap_bytes);
    memcpy(out_host, out.data(), out.size());
    *n_bytes_host = out.size();
    return GF_OK;
}
