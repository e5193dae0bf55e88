// ThreadedRingBuffer<T>: blocking single-producer single-consumer ring used between the OFDM thread and the
// radio thread (/root/reference/src/radio_block.cpp:23-28, :36, :53).  write/read block until everything is
// transferred or the buffer is closed; they return the number of elements transferred.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <thread>
#include <mutex>
#include <vector>
#include "utility/span.h"

template <class T>
class ThreadedRingBuffer {
public:
    explicit ThreadedRingBuffer(size_t capacity) : m_buf(capacity), m_head(0), m_count(0), m_closed(false) {}
    size_t write(tcb::span<const T> src) {
        size_t done = 0;
        std::unique_lock<std::mutex> lock(m_mutex);
        while (done < src.size()) {
            m_cv.wait(lock, [&] { return m_closed || m_count < m_buf.size(); });
            if (m_closed) break;
            const size_t n = std::min(src.size() - done, m_buf.size() - m_count);
            // (at most two contiguous pieces: element-by-element with a modulo per element cost ~1 ms per 230 400-bit frame)
            const size_t at = (m_head + m_count) % m_buf.size(), first = std::min(n, m_buf.size() - at);
            std::copy(src.data() + done, src.data() + done + first, m_buf.data() + at);
            std::copy(src.data() + done + first, src.data() + done + n, m_buf.data());
            m_count += n;
            done += n;
            m_cv.notify_all();
        }
        return done;
    }
    size_t read(tcb::span<T> dst) {
        size_t done = 0;
        std::unique_lock<std::mutex> lock(m_mutex);
        while (done < dst.size()) {
            m_cv.wait(lock, [&] { return m_closed || m_count > 0; });
            if (m_count == 0 && m_closed) break;
            const size_t n = std::min(dst.size() - done, m_count);
            const size_t first = std::min(n, m_buf.size() - m_head);
            std::copy(m_buf.data() + m_head, m_buf.data() + m_head + first, dst.data() + done);
            std::copy(m_buf.data(), m_buf.data() + (n - first), dst.data() + done + first);
            m_head = (m_head + n) % m_buf.size();
            m_count -= n;
            done += n;
            m_cv.notify_all();
        }
        return done;
    }
    void close() {
        std::lock_guard<std::mutex> lock(m_mutex);
        m_closed = true;
        m_cv.notify_all();
    }

private:
    std::vector<T> m_buf;
    size_t m_head, m_count;
    bool m_closed;
    std::mutex m_mutex;
    std::condition_variable m_cv;
};
