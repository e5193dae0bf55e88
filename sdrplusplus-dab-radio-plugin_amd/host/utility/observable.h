// utility/observable.h -- callback list used as `obj.On_X().Attach(fn)` by the reference
// (/root/reference/src/radio_block.cpp:25, :62, :68).
#pragma once
#include <functional>
#include <vector>

template <class... Args>
class Observable {
public:
    using Fn = std::function<void(Args...)>;
    void Attach(Fn fn) { m_fns.push_back(std::move(fn)); }
    void Notify(Args... args) {
        for (auto &f : m_fns) f(args...);
    }
    size_t size() const { return m_fns.size(); }

private:
    std::vector<Fn> m_fns;
};
