// headers.hpp -- the reference's header tokeniser and per-field delta coder (host code).
//
// SURVEY.md 8(f) row 3, first half: read headers are split into fields at every non-alphanumeric
// character, the split is fixed by the first header of the dataset, and every field of every
// following header is coded against the same field of the previous header:
//   STRING  field: one flag byte (0 = same as before, 1 = new value); a new value adds its bytes
//                  to `content` and its length (one byte) to `contentLength`
//   NUMERIC field: the int32 difference to the previous value, four little-endian bytes in `content`
// Mirrors, name for name (including the reference's spelling of HeaderFormatSpeciciation):
//   FieldType, numeric_t, string_t, FIELDLEN_MAX            src/headers.h:12-25
//   HeaderFormatSpeciciation::fromHeader                    src/headers.h:28-41, src/headers.cpp:44-74
//   fromHeader(header, fmt)                                 src/headers.cpp:26-42
//   FieldStorage{,Dst,Src}, store/loadNext{String,Numeric}  src/headers.h:49-103, src/headers.cpp:76-133
// The streams these produce are the INPUT of the reference's libbsc pass (src/workspace.cpp:176-236),
// which stays out of scope (libbsc's source is absent): they are byte-identical to what the
// reference hands to libbsc, not to what it writes into the archive.
// Where the reference has undefined behaviour this file is defined: a numeric field that does
// not fit int32 throws std::invalid_argument (reference: unchecked std::from_chars result),
// differences wrap modulo 2^32 (reference: signed overflow).
#pragma once

#include <cctype>
#include <charconv>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <variant>
#include <vector>

namespace fqcomp28::headers {

constexpr std::size_t FIELDLEN_MAX = 255;

enum class FieldType { NUMERIC = 0, STRING };

using string_t = std::string_view;
using numeric_t = int32_t;
using field_data_t = std::variant<numeric_t, string_t>;
using header_fields_t = std::vector<field_data_t>;

namespace detail {
inline bool alnum(char c) { return std::isalnum(static_cast<unsigned char>(c)) != 0; }
inline bool digit(char c) { return std::isdigit(static_cast<unsigned char>(c)) != 0; }

inline numeric_t parseNumeric(const char *s, const char *e) {
  numeric_t v = 0;
  const auto res = std::from_chars(s, e, v);
  if (res.ec != std::errc())
    throw std::invalid_argument("header field '" + std::string(s, e) + "' is not an int32 number");
  return v;
}

/** end of the field that starts at `s`: the first `sep` strictly behind its first character
 *  (the reference searches from field_start + 1: a field has at least one character) */
inline const char *fieldEnd(const char *s, const char *end, char sep) {
  const char *p = s < end ? s + 1 : end;
  while (p < end && *p != sep) ++p;
  return p;
}
}  // namespace detail

inline field_data_t fieldFromAscii(const char *s, const char *e, FieldType typ) {
  if (typ == FieldType::STRING) return string_t(s, static_cast<std::size_t>(e - s));
  return detail::parseNumeric(s, e);
}

/** Describes the structure of headers: number and types of the fields, separators between them */
struct HeaderFormatSpeciciation {
  std::vector<FieldType> field_types;
  std::vector<char> separators;
  [[nodiscard]] std::size_t n_fields() const { return field_types.size(); }

  /** from an example header: a field is a maximal run of alphanumeric characters; all digits
   *  (or empty) = NUMERIC.  A header that ends in a separator is refused like in the reference. */
  static HeaderFormatSpeciciation fromHeader(std::string_view header) {
    if (header.empty() || header[0] != '@') throw std::invalid_argument("header should start with '@'");
    HeaderFormatSpeciciation fmt;
    const char *p = header.data() + 1, *const end = header.data() + header.size();
    for (;;) {
      bool numeric = true;
      for (; p < end && detail::alnum(*p); ++p) numeric = numeric && detail::digit(*p);
      fmt.field_types.push_back(numeric ? FieldType::NUMERIC : FieldType::STRING);
      if (p == end) break;
      if (p == end - 1) throw std::invalid_argument(std::string(header) + ": header should end in alnum char");
      fmt.separators.push_back(*p++);
    }
    return fmt;
  }
  friend bool operator==(const HeaderFormatSpeciciation &a, const HeaderFormatSpeciciation &b) {
    return a.field_types == b.field_types && a.separators == b.separators;
  }
};
using HeaderFormatSpecification = HeaderFormatSpeciciation;

/** Header fields of `header` according to `fmt` (string fields point into `header`) */
inline header_fields_t fromHeader(std::string_view header, const HeaderFormatSpeciciation &fmt) {
  header_fields_t fields(fmt.n_fields());
  const char *s = header.data() + 1, *const end = header.data() + header.size();
  for (std::size_t i = 0; i + 1 < fmt.n_fields(); ++i) {
    const char *e = detail::fieldEnd(s, end, fmt.separators[i]);
    fields[i] = fieldFromAscii(s, e, fmt.field_types[i]);
    s = e < end ? e + 1 : end;
  }
  if (!fields.empty()) fields.back() = fieldFromAscii(s, end, fmt.field_types.back());
  return fields;
}

/** One field of many headers */
struct FieldStorage {
  std::vector<std::byte> isDifferentFlag, content, contentLength;
  struct sizes {  // original sizes, as the archive records them per field
    uint32_t isDifferentFlag = 0, content = 0, contentLength = 0;
    friend bool operator==(const sizes &a, const sizes &b) {
      return a.isDifferentFlag == b.isDifferentFlag && a.content == b.content && a.contentLength == b.contentLength;
    }
  };
  [[nodiscard]] sizes originalSizes() const {
    return {static_cast<uint32_t>(isDifferentFlag.size()), static_cast<uint32_t>(content.size()),
            static_cast<uint32_t>(contentLength.size())};
  }
  friend bool operator==(const FieldStorage &a, const FieldStorage &b) {
    return a.isDifferentFlag == b.isDifferentFlag && a.content == b.content && a.contentLength == b.contentLength;
  }
  virtual ~FieldStorage() = default;
  virtual void clear() { isDifferentFlag.clear(); content.clear(); contentLength.clear(); }
};

struct FieldStorageDst : FieldStorage {
  void storeString(const char *field_start, const char *field_end, string_t &prev_val) {
    const string_t val(field_start, static_cast<std::size_t>(field_end - field_start));
    if (val == prev_val) {
      isDifferentFlag.push_back(std::byte{0});
      return;
    }
    if (val.size() >= FIELDLEN_MAX) throw std::invalid_argument("header field longer than 254 characters");
    isDifferentFlag.push_back(std::byte{1});
    const auto *b = reinterpret_cast<const std::byte *>(val.data());
    content.insert(content.end(), b, b + val.size());
    contentLength.push_back(static_cast<std::byte>(val.size()));
    prev_val = val;
  }
  void storeNumeric(const char *field_start, const char *field_end, numeric_t &prev_val) {
    const numeric_t val = detail::parseNumeric(field_start, field_end);
    const uint32_t delta = static_cast<uint32_t>(val) - static_cast<uint32_t>(prev_val);
    for (int i = 0; i < 4; ++i) content.push_back(static_cast<std::byte>((delta >> (8 * i)) & 0xFFu));
    prev_val = val;
  }
};

struct FieldStorageSrc : FieldStorage {
  struct { std::size_t isDifferentPos = 0, contentPos = 0, contentLengthPos = 0; } index;

  /** @return number of bytes written to dst (which must have FIELDLEN_MAX bytes of room) */
  unsigned loadNextString(char *dst, string_t &prev_val) {
    if (index.isDifferentPos >= isDifferentFlag.size()) throw std::out_of_range("header field flags exhausted");
    if (isDifferentFlag[index.isDifferentPos++] == std::byte{0}) {
      std::memmove(dst, prev_val.data(), prev_val.size());
      return static_cast<unsigned>(prev_val.size());
    }
    if (index.contentLengthPos >= contentLength.size()) throw std::out_of_range("header field lengths exhausted");
    const unsigned len = static_cast<unsigned char>(contentLength[index.contentLengthPos++]);
    if (index.contentPos + len > content.size()) throw std::out_of_range("header field content exhausted");
    std::memcpy(dst, content.data() + index.contentPos, len);
    index.contentPos += len;
    prev_val = string_t(dst, len);
    return len;
  }
  unsigned loadNextNumeric(char *dst, numeric_t &prev_val) {
    if (index.contentPos + 4 > content.size()) throw std::out_of_range("header field content exhausted");
    uint32_t delta = 0;
    for (int i = 0; i < 4; ++i) delta |= static_cast<uint32_t>(content[index.contentPos + i]) << (8 * i);
    index.contentPos += 4;
    const numeric_t val = static_cast<numeric_t>(static_cast<uint32_t>(prev_val) + delta);
    prev_val = val;
    const auto res = std::to_chars(dst, dst + FIELDLEN_MAX, val);
    return static_cast<unsigned>(res.ptr - dst);
  }
  void clear() override { FieldStorage::clear(); index = {}; }
};

using CompressedFieldStorage = FieldStorage;

/** One header through the field coders (CompressionWorkspace::encodeHeader, src/workspace.cpp:95-126) */
inline void encodeHeader(std::string_view header, const HeaderFormatSpeciciation &fmt, header_fields_t &prev,
                         std::vector<FieldStorageDst> &fields) {
  const char *s = header.data() + 1, *const end = header.data() + header.size();
  for (std::size_t i = 0, n = fmt.n_fields(); i < n; ++i) {
    const char *e = i + 1 < n ? detail::fieldEnd(s, end, fmt.separators[i]) : end;
    if (fmt.field_types[i] == FieldType::STRING)
      fields[i].storeString(s, e, std::get<string_t>(prev[i]));
    else
      fields[i].storeNumeric(s, e, std::get<numeric_t>(prev[i]));
    s = e < end ? e + 1 : end;
  }
}

/** The next header, written to dst ('@' included); returns its length
 *  (DecompressionWorkspace::decodeHeader, src/workspace.cpp:128-157) */
inline unsigned decodeHeader(char *dst, const HeaderFormatSpeciciation &fmt, header_fields_t &prev,
                             std::vector<FieldStorageSrc> &fields) {
  char *const start = dst;
  *dst++ = '@';
  for (std::size_t i = 0, n = fmt.n_fields(); i < n; ++i) {
    if (fmt.field_types[i] == FieldType::STRING)
      dst += fields[i].loadNextString(dst, std::get<string_t>(prev[i]));
    else
      dst += fields[i].loadNextNumeric(dst, std::get<numeric_t>(prev[i]));
    if (i + 1 < n) *dst++ = fmt.separators[i];
  }
  return static_cast<unsigned>(dst - start);
}

}  // namespace fqcomp28::headers
