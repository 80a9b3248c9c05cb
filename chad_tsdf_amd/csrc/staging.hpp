// staging.hpp -- host side of chad_map_insert (the reference's signature: points in pageable host memory, borrowed for the call,
// tsdf.hpp:50-57): the frame is copied into the slot's pinned staging buffer chunk by chunk, every chunk's DMA starting as soon as
// the chunk is in place.  One thread copies at ~25 GB/s and the DMA moves 30-50: a helper thread (sleeping between frames) shares
// the copy, the calling thread copies too and issues the DMAs in order (0.57 -> 0.40 ms per 12 MB frame; more helpers gain nothing).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace chadhip {

class CopyPool {
public:
    explicit CopyPool(int helpers) {
        for (int k = 0; k < helpers; k++) workers_.emplace_back([this] { work(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lock(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // Copies src[0, bytes) to dst in chunks; ready(offset, length) is called on the CALLING thread for every chunk, in order, once
    // the chunk is in place (it starts the chunk's DMA).  Returns when everything has been copied and reported.
    template <class Ready>
    void copy(const char* src, char* dst, size_t bytes, size_t chunk, Ready ready) {
        const size_t chunks = (bytes + chunk - 1) / chunk;
        if (done_.size() < chunks) done_ = std::vector<std::atomic<uint8_t>>(chunks);
        for (size_t c = 0; c < chunks; c++) done_[c].store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lock(mu_);
            src_ = src; dst_ = dst; bytes_ = bytes; chunk_ = chunk; chunks_ = chunks;
            next_.store(0, std::memory_order_relaxed);
            busy_.store(int(workers_.size()), std::memory_order_relaxed);
            generation_++;
        }
        cv_.notify_all();
        size_t issued = 0;
        while (issued < chunks) {
            if (done_[issued].load(std::memory_order_acquire)) {
                const size_t at = issued * chunk;
                ready(at, std::min(chunk, bytes - at));
                issued++;
                continue;
            }
            if (!copy_one()) std::this_thread::yield();   // nothing left to grab: the chunk awaited is in another thread's hands
        }
        while (busy_.load(std::memory_order_acquire) != 0) std::this_thread::yield();   // no helper still looks at this job
    }

private:
    bool copy_one() {
        const size_t c = next_.fetch_add(1, std::memory_order_relaxed);
        if (c >= chunks_) return false;
        const size_t at = c * chunk_;
        std::memcpy(dst_ + at, src_ + at, std::min(chunk_, bytes_ - at));
        done_[c].store(1, std::memory_order_release);
        return true;
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            while (copy_one()) {}
            busy_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
    uint64_t generation_ = 0;
    const char* src_ = nullptr;
    char* dst_ = nullptr;
    size_t bytes_ = 0, chunk_ = 0, chunks_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<int> busy_{0};
    std::vector<std::atomic<uint8_t>> done_;
};

}  // namespace chadhip
