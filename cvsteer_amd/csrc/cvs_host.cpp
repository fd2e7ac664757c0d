// cvs_host.cpp -- host planes in, host planes out: upload, filtering and download overlapped band by band.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvs_context.h"

namespace cvs {

// Overlapped host path.  The reference's callers hand over HOST images and expect HOST results
// (test/test.cpp:73,85-90; example/steer.cpp:73-104).  Done naively that is upload, kernel, download, one after the
// other: 64 MiB up + 128 MiB down at 56 GB/s each = 3.6 ms around a 0.11 ms kernel.  The host link is full duplex, so
// the image is cut into row bands and three things run at once: the upload of band b+1 (this thread, stream s_up), the
// filtering of band b (the handle's stream; cvs_setup_rows machinery, values bit-identical to a whole-image launch)
// and the download of band b-1's outputs (a second host thread, stream s_down).  Host memory may be pageable: the
// runtime pins it on the fly (tools/pcie_probe.hip: pageable = pinned = 56 GB/s per direction; both directions from
// two threads 2.66 ms instead of 3.58).  What remains is max(upload, download) plus one band of latency.
int host_pipeline(cvs_handle h, Call& c, BasisArgs& a, float* scr)
{
    const int W = h->width;
    int nbands = 8;
    int per = (a.rows + nbands - 1) / nbands;
    per = std::max(a.strip_rows, (per + a.strip_rows - 1) / a.strip_rows * a.strip_rows);
    nbands = (a.rows + per - 1) / per;
    if (!h->s_up) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_down, hipStreamNonBlocking));
    }
    while ((int)h->band_ev.size() < 2 * nbands + 1) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->band_ev.push_back(e);
    }
    hipEvent_t* up = h->band_ev.data();
    hipEvent_t* comp = h->band_ev.data() + nbands;
    // the copy streams start behind whatever the handle's stream still has queued on these buffers
    hipEvent_t start = h->band_ev[2 * nbands];
    HIP_TRY(h, hipEventRecord(start, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->s_up, start, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->s_down, start, 0));
    // plain order: the launch tuner works on whole images, not on bands
    a.block_order = 0;
    note_launch(h, a);

    // download thread: band b's outputs leave as soon as its kernel has finished
    std::mutex mu;
    std::condition_variable cv;
    int enqueued = 0;
    bool abort_dl = false;
    hipError_t dl_err = hipSuccess;
    const std::vector<Pending> outs = c.outs;
    const int rows = a.rows, device = h->device;
    hipStream_t s_down = h->s_down;
    std::thread downloader;
    if (!outs.empty()) {
        downloader = std::thread([&, rows, device, s_down, per, nbands] {
            hipError_t e = hipSetDevice(device);
            for (int b = 0; b < nbands && e == hipSuccess; ++b) {
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return enqueued > b || abort_dl; });
                    if (abort_dl) break;
                }
                const int lo = b * per, hi = std::min(rows, lo + per);
                e = hipStreamWaitEvent(s_down, comp[b], 0);
                for (const Pending& o : outs) {
                    if (e != hipSuccess) break;
                    if (o.host->step == o.pitch * sizeof(float) && o.host->step == (size_t)o.host->cols * sizeof(float)) {  // dense on both sides
                        e = hipMemcpyAsync(reinterpret_cast<char*>(o.host->data) + (size_t)lo * o.host->step, o.dev + (size_t)lo * o.pitch,
                                           (size_t)(hi - lo) * o.host->step, hipMemcpyDeviceToHost, s_down);
                        continue;
                    }
                    e = copy_rows(reinterpret_cast<char*>(o.host->data) + (size_t)lo * o.host->step, o.host->step, o.dev + (size_t)lo * o.pitch,
                                         o.pitch * sizeof(float), (size_t)o.host->cols * sizeof(float), hi - lo, hipMemcpyDeviceToHost, s_down);
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s_down);
            dl_err = e;
        });
    }
    auto stop = [&](int rc) {
        {
            std::lock_guard<std::mutex> lock(mu);
            abort_dl = true;
        }
        cv.notify_all();
        if (downloader.joinable()) downloader.join();
        return rc;
    };
    const cvs_plane* img = c.deferred_image;  // nullptr: the image is already on the device, only outputs travel
    int up_to = 0;                            // rows of the image uploaded so far
    for (int b = 0; b < nbands; ++b) {
        const int lo = b * per, hi = std::min(a.rows, lo + per);
        if (img) {
            const int need = std::min(a.rows, hi + W);  // the band's kernel reads W rows beyond its last output row
            if (need > up_to) {
                hipError_t e;
                if (c.deferred_u8) {
                    e = copy_rows(c.deferred_u8 + (size_t)up_to * c.deferred_u8_pitch, c.deferred_u8_pitch,
                                         reinterpret_cast<const char*>(img->data) + (size_t)up_to * img->step, img->step, (size_t)img->cols, need - up_to,
                                         hipMemcpyHostToDevice, h->s_up);
                    if (e == hipSuccess)
                        e = launch_u8_to_f32(c.deferred_u8 + (size_t)up_to * c.deferred_u8_pitch, c.deferred_u8_pitch, need - up_to, img->cols,
                                             const_cast<float*>(a.in) + (size_t)up_to * a.in_pitch, a.in_pitch, h->s_up);
                } else {
                    e = copy_rows(const_cast<float*>(a.in) + (size_t)up_to * a.in_pitch, a.in_pitch * sizeof(float),
                                         reinterpret_cast<const char*>(img->data) + (size_t)up_to * img->step, img->step, (size_t)img->cols * sizeof(float),
                                         need - up_to, hipMemcpyHostToDevice, h->s_up);
                }
                if (e != hipSuccess) return stop(fail_hip(h, e, "host pipeline upload"));
                up_to = need;
            }
            hipError_t e = hipEventRecord(up[b], h->s_up);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, up[b], 0);
            if (e != hipSuccess) return stop(fail_hip(h, e, "host pipeline ordering"));
        }
        BasisArgs ab = a;
        ab.out_row_lo = lo;
        ab.out_row_hi = hi;
        hipError_t e = launch_basis(h->kind, h->width, h->taps, ab, scr, h->stream);
        if (e == hipSuccess) e = hipEventRecord(comp[b], h->stream);
        if (e != hipSuccess) return stop(fail_hip(h, e, "host pipeline launch"));
        {
            std::lock_guard<std::mutex> lock(mu);
            enqueued = b + 1;
        }
        cv.notify_all();
    }
    if (downloader.joinable()) downloader.join();
    if (dl_err != hipSuccess) return fail_hip(h, dl_err, "host pipeline download");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    c.outs.clear();  // nothing left for finish() to copy
    return CVS_OK;
}

}  // namespace cvs
