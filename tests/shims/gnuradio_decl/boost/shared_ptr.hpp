// declaration-only stand-in (see ../README.md)
#pragma once
namespace boost {
template <class T> class shared_ptr {
public:
    shared_ptr();
    template <class Y> explicit shared_ptr(Y *p);
    template <class Y> shared_ptr(const shared_ptr<Y> &r);
    T *operator->() const;
    T &operator*() const;
    T *get() const;
    explicit operator bool() const;
};
}  // namespace boost
