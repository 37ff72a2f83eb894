// sdm_io/binary_archive.hpp -- reader/writer for the reference's on-disk model format.
//
// The reference serialises models with cereal 1.1.1's portable-less BinaryOutputArchive
// (include/rcr/model.hpp:192-219, 3rdparty/cereal-1.1.1/include/cereal/archives/binary.hpp:51-155): raw
// little-endian bytes, no padding, no class-version fields; std::vector / std::string are a u64 element
// count followed by the elements (cereal/types/vector.hpp:65-70, string.hpp); enums are written as their
// underlying type (cereal/types/common.hpp:91-98); cv::Mat is `i32 rows, i32 cols, i32 type, u8 continuous,
// raw bytes` (include/superviseddescent/utils/mat_cerealisation.hpp:42-67).  These two archive classes speak
// exactly that layout through the same `ar(a, b, c)` call syntax, so the serialize() members of the header
// layer read like the reference's and files are interchangeable with it.
#pragma once

#ifndef SDM_IO_BINARY_ARCHIVE_HPP_
#define SDM_IO_BINARY_ARCHIVE_HPP_

#include "sdm_cv/core.hpp"

#include <cstdint>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace sdm_io {

struct Exception : public std::runtime_error {   // stands in for cereal::Exception (binary.hpp:100-101)
    explicit Exception(const std::string& w) : std::runtime_error(w) {}
};

class BinaryOutputArchive {
public:
    explicit BinaryOutputArchive(std::ostream& os) : os_(os) {}
    template <class... T> void operator()(T&&... v) { (save(v), ...); }
    void write(const void* p, size_t n)
    {
        os_.write((const char*)p, (std::streamsize)n);
        if (!os_) throw Exception("Failed to write " + std::to_string(n) + " bytes to output stream");
    }

private:
    template <class T> typename std::enable_if<std::is_arithmetic<T>::value>::type save(const T& v) { write(&v, sizeof(T)); }
    template <class T> typename std::enable_if<std::is_enum<T>::value>::type save(const T& v)
    {
        const typename std::underlying_type<T>::type u = static_cast<typename std::underlying_type<T>::type>(v);
        write(&u, sizeof(u));
    }
    void save(const std::string& s)
    {
        const uint64_t n = s.size();
        write(&n, 8);
        write(s.data(), s.size());
    }
    void save(const cv::Mat& m)
    {
        const int rows = m.rows, cols = m.cols, type = m.type();
        const bool continuous = m.isContinuous();
        (*this)(rows, cols, type, continuous);
        const size_t row_bytes = (size_t)cols * m.elemSize();
        if (continuous) { if (rows > 0) write(m.ptr<uint8_t>(0), row_bytes * (size_t)rows); }
        else for (int r = 0; r < rows; ++r) write(m.ptr<uint8_t>(r), row_bytes);
    }
    template <class T> void save(const std::vector<T>& v)
    {
        const uint64_t n = v.size();
        write(&n, 8);
        for (const auto& e : v) save(e);
    }
    template <class T>
    typename std::enable_if<std::is_class<T>::value && !std::is_same<T, std::string>::value && !std::is_same<T, cv::Mat>::value>::type
    save(const T& v)
    {
        const_cast<T&>(v).serialize(*this);
    }
    std::ostream& os_;
};

class BinaryInputArchive {
public:
    explicit BinaryInputArchive(std::istream& is) : is_(is) {}
    template <class... T> void operator()(T&&... v) { (load(v), ...); }
    void read(void* p, size_t n)
    {
        is_.read((char*)p, (std::streamsize)n);
        if ((size_t)is_.gcount() != n)
            throw Exception("Failed to read " + std::to_string(n) + " bytes from input stream! Read " + std::to_string(is_.gcount()));
    }

private:
    template <class T> typename std::enable_if<std::is_arithmetic<T>::value>::type load(T& v) { read(&v, sizeof(T)); }
    template <class T> typename std::enable_if<std::is_enum<T>::value>::type load(T& v)
    {
        typename std::underlying_type<T>::type u;
        read(&u, sizeof(u));
        v = static_cast<T>(u);
    }
    void load(std::string& s)
    {
        uint64_t n = 0;
        read(&n, 8);
        if (n > (1ull << 32)) throw Exception("implausible string length in archive");
        s.resize((size_t)n);
        if (n) read(&s[0], (size_t)n);
    }
    void load(cv::Mat& m)
    {
        int rows = 0, cols = 0, type = 0;
        bool continuous = true;
        (*this)(rows, cols, type, continuous);
        if (rows < 0 || cols < 0 || (type != CV_32FC1 && type != CV_8UC1 && type != CV_8UC3))
            throw Exception("unsupported cv::Mat header in archive");
        m.create(rows, cols, type);
        const size_t row_bytes = (size_t)cols * m.elemSize();
        for (int r = 0; r < rows; ++r) read(m.ptr<uint8_t>(r), row_bytes);
    }
    template <class T> void load(std::vector<T>& v)
    {
        uint64_t n = 0;
        read(&n, 8);
        if (n > (1ull << 32)) throw Exception("implausible vector length in archive");
        v.resize((size_t)n);
        for (auto& e : v) load(e);
    }
    template <class T>
    typename std::enable_if<std::is_class<T>::value && !std::is_same<T, std::string>::value && !std::is_same<T, cv::Mat>::value>::type
    load(T& v)
    {
        v.serialize(*this);
    }
    std::istream& is_;
};

}  // namespace sdm_io
#endif
