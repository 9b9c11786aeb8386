// Host side of the JPEG ingest stage: marker parsing and Huffman entropy
// decoding of baseline JPEG streams into SPARSE quantised coefficients.  Everything after the
// entropy decoder (dequantisation, inverse DCT, chroma upsampling, colour conversion) runs on the
// GPU (k_jpeg.hip), so what crosses PCIe is the list of non-zero coefficients (~0.13 MB for a
// 640x480 camera frame) instead of 0.92 MB of decoded pixels.
//
// Replaces the decode half of duckietown_utils.jpg.image_cv_from_jpg = cv2.imdecode(data,
// IMREAD_COLOR) (ref: src/duckietown/include/duckietown_utils/jpg.py:21-31), i.e. libjpeg-turbo's
// default decoder (ITU-T T.81 baseline sequential Huffman).
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lf {
namespace jpeg {

// One per frame, copied to the device as is.
struct FrameHeader {
    int32_t valid;          // 0: the stream could not be decoded, the frame is written as zeros
    int32_t ncomp;          // 1 (grayscale) or 3
    int32_t hmax, vmax;     // luma sampling factors (chroma is 1x1): 1x1, 2x1 or 2x2
    int32_t mcux, mcuy;     // MCUs per row / column
    int32_t is_rgb;         // components are R,G,B (Adobe transform 0), not Y,Cb,Cr
    int32_t nblocks;        // 8x8 blocks in the scan, MCU order
    uint32_t entry_base;    // first entry of this frame in the batch's entry array
    uint32_t block_base;    // first block of this frame in the batch's block_end array
    uint16_t qt[3][64];     // quantisation table per component, natural (row-major) order
};

struct FrameCoefs {
    FrameHeader hdr;
    int rows, cols;
    int status;                         // lf_status of this frame
    std::vector<uint32_t> entries;      // (natural position << 16) | (uint16_t)quantised value, block after block;
                                        // the vector is only ever grown (capacity), n_entries of it are valid
    size_t n_entries;
    std::vector<uint32_t> block_end;    // entries of blocks 0..b, per block (hdr.nblocks valid)
};

// ---- entropy decoding ON THE DEVICE (k_jhuff.hip): the host only parses the headers.
// One canonical Huffman table as both decoders use it: 9-bit lookahead ((length << 8) | symbol, 0 = longer code), then
// maxcode / delta per length (T.81 Annex C / F.2.2.3).
struct HuffDev {
    uint16_t fast[512];
    int32_t maxcode[18];
    int32_t delta[17];
    int32_t present;
    uint8_t vals[256];
};
// Per frame, copied to the device as is.
struct DevFrame {
    FrameHeader hdr;            // valid = 1 when the headers parsed; the device clears it when the entropy data is corrupt
    uint32_t scan_off;          // entropy-coded data of this frame in the batch's byte buffer (raw: stuffed, with markers)
    uint32_t scan_len;          // bytes from the end of the SOS header to the end of the stream
    int32_t restart;            // MCUs per restart interval, 0 = none
    int32_t n_mcu, bpm, luma;   // MCUs in the scan, blocks per MCU, luma blocks per MCU
    int32_t tab_dc[3], tab_ac[3];   // per component: its DC table (0..3) and AC table (4..7) in tabs[]
    uint32_t coef_base;         // first block of this frame in the batch's dense coefficient array
    uint32_t clean_off;         // this frame's region of the unstuffed byte buffer / of the per-subsequence arrays
    uint32_t sub_off, seg_off;
    int32_t max_sub, max_seg;   // capacity of those regions
    HuffDev tabs[8];
};
// Headers only: fills out (tables, geometry, where the scan starts); lf_status.  scan_begin = offset of the entropy data
// in `data`.
int prepare_device_frame(const uint8_t* data, size_t size, int expect_rows, int expect_cols, DevFrame& out, size_t* scan_begin);

// rows / cols / components / sampling of a stream; lf_status.
int peek(const uint8_t* data, size_t size, int* rows, int* cols, int* ncomp, int* hmax, int* vmax);

// Parse and entropy-decode one stream.  Returns the frame's lf_status (also stored in out.status);
// on failure out.hdr.valid = 0 and the coefficient lists are empty.  expect_rows / expect_cols > 0: a stream of
// any other size is refused (LF_ERR_BAD_ARG) right after its headers, before any buffer is sized from them.
// Host allocations are bounded by the stream's length; allocation failure is a status (LF_ERR_CAPACITY), never
// an exception.
int decode_coefficients(const uint8_t* data, size_t size, FrameCoefs& out, int expect_rows = 0, int expect_cols = 0);

// Persistent host threads for the per-frame work (entropy decoding, packing the staging buffer):
// run(n, job) executes job(worker) on n workers and returns when all are done.  Threads are created
// once and parked on a condition variable between calls.
class WorkerPool {
public:
    WorkerPool() = default;
    ~WorkerPool();
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    void run(int n_workers, const std::function<void(int)>& job);

private:
    void loop(int id);
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_, done_;
    const std::function<void(int)>* job_ = nullptr;
    unsigned long generation_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

}  // namespace jpeg
}  // namespace lf
