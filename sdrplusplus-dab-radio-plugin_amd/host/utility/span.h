// utility/span.h -- minimal tcb::span (the reference includes "utility/span.h",
// /root/reference/src/radio_block.cpp:7, src/dab_module.cpp:16).  Borrowed view: pointer + length.
#pragma once
#include <cstddef>
#include <type_traits>
#include <vector>

namespace tcb {

template <class T>
class span {
public:
    using element_type = T;
    using value_type = typename std::remove_cv<T>::type;
    constexpr span() noexcept : m_data(nullptr), m_size(0) {}
    constexpr span(T *data, size_t size) noexcept : m_data(data), m_size(size) {}
    template <class U, class = typename std::enable_if<std::is_convertible<U (*)[], T (*)[]>::value>::type>
    constexpr span(const span<U> &o) noexcept : m_data(o.data()), m_size(o.size()) {}
    template <class A>
    span(std::vector<value_type, A> &v) noexcept : m_data(v.data()), m_size(v.size()) {}
    template <class A, class Q = T, class = typename std::enable_if<std::is_const<Q>::value>::type>
    span(const std::vector<value_type, A> &v) noexcept : m_data(v.data()), m_size(v.size()) {}
    template <size_t N>
    constexpr span(T (&arr)[N]) noexcept : m_data(arr), m_size(N) {}

    constexpr T *data() const noexcept { return m_data; }
    constexpr size_t size() const noexcept { return m_size; }
    constexpr bool empty() const noexcept { return m_size == 0; }
    constexpr T &operator[](size_t i) const { return m_data[i]; }
    constexpr T *begin() const noexcept { return m_data; }
    constexpr T *end() const noexcept { return m_data + m_size; }
    constexpr span first(size_t n) const { return span(m_data, n); }
    constexpr span subspan(size_t off) const { return span(m_data + off, m_size - off); }
    constexpr span subspan(size_t off, size_t n) const { return span(m_data + off, n); }

private:
    T *m_data;
    size_t m_size;
};

template <class T> span(T *, size_t) -> span<T>;
template <class T, class A> span(std::vector<T, A> &) -> span<T>;
template <class T, class A> span(const std::vector<T, A> &) -> span<const T>;

}  // namespace tcb
