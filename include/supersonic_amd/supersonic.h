// supersonic_amd/supersonic.h -- C++ host-side mirror of the builder API of
// supersonic/supersonic.h for the Filter -> Project/Compute -> Aggregate (+Sort) path,
// implemented over the C ABI of include/ssgpu.h (libssgpu.so).  Header-only.
//
// Same names, argument meaning, ownership and error behaviour as the reference:
//   * factories take OWNERSHIP of the raw pointers they are given
//     (cursor/core/compute.h, filter.h, aggregate.h:224-342, sort.h:83-86);
//   * Operation::CreateCursor() binds the whole tree and returns FailureOrOwned<Cursor>
//     (cursor/base/operation.h:62) -- bind errors carry the reference's ReturnCode;
//   * Cursor::Next(max_row_count) returns a ResultView whose View stays valid until the
//     next call (cursor/base/cursor.h:131-148); Interrupt() may come from another thread.
// Execution differs by design: the first Next() runs the whole fused pipeline on the GPU,
// later calls slice the finished result.  There is no CPU execution path.
#ifndef SUPERSONIC_AMD_SUPERSONIC_H_
#define SUPERSONIC_AMD_SUPERSONIC_H_

#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include <stdio.h>
#include <stdlib.h>

#include <deque>
#include <map>
#include <ostream>
#include <set>

#include "../ssgpu.h"

// The reference's integral types (supersonic/utils/integral_types.h:22-43) live in the global namespace, and user code
// written against supersonic/supersonic.h uses them unqualified (test/guide/primer.cc: `int32 a[8]`, `const int32* result`).
#ifndef SUPERSONIC_AMD_NO_GLOBAL_INTEGRAL_TYPES
typedef signed char int8;
typedef short int16;
typedef int int32;
typedef long long int64;
typedef unsigned char uint8;
typedef unsigned short uint16;
typedef unsigned int uint32;
typedef unsigned long long uint64;
#endif

// The reference keeps StringPiece, `string` and its map helpers in the GLOBAL namespace (supersonic/utils/strings/stringpiece.h,
// supersonic/utils/map_util.h), and its guides use them unqualified (test/guide/group_sort.cc:170-209: `const StringPiece*`,
// `string names_str[...]`, `ContainsKey(max_age, key)`); `supersonic::StringPiece` names the same class.
using std::string;

// STRING cells of a View (supersonic/utils/strings/stringpiece.h): a pointer + length, not owning the bytes.  Views handed to
// ScanView / Evaluate hold arrays of these; across the C ABI they travel as INT32 codes of an order-preserving
// dictionary (ssgpu_dict_*), and result Views point into the cursor's dictionary.
class StringPiece {
 public:
  StringPiece() : data_(""), length_(0) {}
  StringPiece(const char* s) : data_(s), length_(s ? strlen(s) : 0) {}            // NOLINT(runtime/explicit)
  StringPiece(const std::string& s) : data_(s.data()), length_(s.size()) {}       // NOLINT(runtime/explicit)
  StringPiece(const char* s, size_t n) : data_(s), length_(n) {}
  const char* data() const { return data_; }
  size_t size() const { return length_; }
  size_t length() const { return length_; }
  bool empty() const { return length_ == 0; }
  std::string ToString() const { return std::string(data_, length_); }
  int compare(const StringPiece& o) const {
    const int r = memcmp(data_, o.data_, std::min(length_, o.length_));
    return r != 0 ? r : (length_ < o.length_ ? -1 : (length_ > o.length_ ? 1 : 0));
  }
 private:
  const char* data_;
  size_t length_;
};
inline bool operator==(const StringPiece& a, const StringPiece& b) { return a.size() == b.size() && memcmp(a.data(), b.data(), a.size()) == 0; }
inline bool operator!=(const StringPiece& a, const StringPiece& b) { return !(a == b); }
inline bool operator<(const StringPiece& a, const StringPiece& b) { return a.compare(b) < 0; }

inline std::ostream& operator<<(std::ostream& o, const StringPiece& piece) { o.write(piece.data(), static_cast<std::streamsize>(piece.size())); return o; }

// supersonic/utils/map_util.h:100-330 (the helpers user code of this path calls)
template <class Collection, class Key> bool ContainsKey(const Collection& collection, const Key& key) { return collection.find(key) != collection.end(); }
template <class Collection, class Key, class Value> bool ContainsKeyValuePair(const Collection& collection, const Key& key, const Value& value) {
  auto range = collection.equal_range(key);
  for (auto it = range.first; it != range.second; ++it) if (it->second == value) return true;
  return false;
}
template <class Collection> const typename Collection::value_type::second_type* FindOrNull(const Collection& collection, const typename Collection::value_type::first_type& key) {
  auto it = collection.find(key);
  return it == collection.end() ? nullptr : &it->second;
}
template <class Collection> const typename Collection::value_type::second_type& FindWithDefault(const Collection& collection, const typename Collection::value_type::first_type& key,
                                                                                                const typename Collection::value_type::second_type& value) {
  auto it = collection.find(key);
  return it == collection.end() ? value : it->second;
}
template <class Collection> bool InsertIfNotPresent(Collection* const collection, const typename Collection::value_type::first_type& key, const typename Collection::value_type::second_type& value) {
  return collection->insert(typename Collection::value_type(key, value)).second;
}
template <class Collection> bool InsertOrUpdate(Collection* const collection, const typename Collection::value_type::first_type& key, const typename Collection::value_type::second_type& value) {
  auto ret = collection->insert(typename Collection::value_type(key, value));
  if (!ret.second) { ret.first->second = value; return false; }
  return true;
}

namespace supersonic {

using std::string;
using ::StringPiece;

// base/infrastructure/types.h:252-256: row counts are UNSIGNED (so Next(-1) asks for "as many rows as you have"), row ids signed
typedef unsigned long long rowcount_t;
typedef long long rowid_t;

// ---- enums: numeric values of supersonic/proto/supersonic.proto ------------------------
enum DataType { INT32 = 1, INT64 = 2, UINT64 = 3, DATETIME = 4, DOUBLE = 5, BOOL = 6, UINT32 = 8, FLOAT = 9, DATE = 10, STRING = 0, BINARY = 7 };
enum Nullability { NOT_NULLABLE = 0, NULLABLE = 1 };
enum Aggregation { SUM = 0, MIN = 1, MAX = 2, COUNT = 3, CONCAT = 4, FIRST = 5, LAST = 6 };
enum ColumnOrder { ASCENDING = 0, DESCENDING = 1 };
enum JoinType { INNER = 0, LEFT_OUTER = 1 };          // supersonic.proto:108-113 (RIGHT/FULL_OUTER: not on device)
enum KeyUniqueness { NOT_UNIQUE = 0, UNIQUE = 1 };    // supersonic.proto:115-118
enum ReturnCode {
  OK = 0, ERROR_UNKNOWN_ERROR = 100, ERROR_GENERAL_IO_ERROR = 101, ERROR_MEMORY_EXCEEDED = 102, ERROR_NOT_IMPLEMENTED = 103,
  ERROR_EVALUATION_ERROR = 104, ERROR_TOO_MANY_ROWS = 302, ERROR_ATTRIBUTE_COUNT_MISMATCH = 401,
  ERROR_ATTRIBUTE_TYPE_MISMATCH = 402, ERROR_ATTRIBUTE_MISSING = 403, ERROR_ATTRIBUTE_EXISTS = 404,
  ERROR_INVALID_ARGUMENT_TYPE = 405, ERROR_INVALID_ARGUMENT_VALUE = 407, INTERRUPTED = 1000
};

// ---- errors (base/exception/exception.h:53, result.h:43-130) ----------------------------
class Exception {
 public:
  Exception(int code, const std::string& message) : code_(code), message_(message) {}
  ReturnCode return_code() const { return static_cast<ReturnCode>(code_); }
  const std::string& message() const { return message_; }
  // exception.h: the message with its code (there are no stack frames to print on this side of the C ABI)
  std::string ToString() const { return "Exception " + std::to_string(code_) + ": " + message_; }
  std::string PrintStackTrace() const { return ToString(); }
 private:
  int code_;
  std::string message_;
};

template <typename T>
class FailureOrOwned {
 public:
  explicit FailureOrOwned(T* value) : value_(value) {}
  explicit FailureOrOwned(Exception* e) : exception_(e) {}
  FailureOrOwned(FailureOrOwned&& o) = default;
  bool is_failure() const { return exception_ != nullptr; }
  bool is_success() const { return !is_failure(); }
  const Exception& exception() const { return *exception_; }
  T* get() const { return value_.get(); }
  T* release() { return value_.release(); }
  T* operator->() const { return value_.get(); }
  T& operator*() const { return *value_; }
  Exception* release_exception() { return exception_.release(); }
 private:
  std::unique_ptr<T> value_;
  std::unique_ptr<Exception> exception_;
};

// FailureOr<T> / FailureOrVoid (utils/exception/failureor.h): a value or an owned Exception.
template <typename T>
class FailureOr {
 public:
  explicit FailureOr(const T& value) : value_(value) {}
  explicit FailureOr(Exception* e) : value_(), exception_(e) {}
  FailureOr(FailureOr&& o) = default;
  bool is_failure() const { return exception_ != nullptr; }
  bool is_success() const { return !is_failure(); }
  const Exception& exception() const { return *exception_; }
  const T& get() const { return value_; }
  Exception* release_exception() { return exception_.release(); }
 private:
  T value_;
  std::unique_ptr<Exception> exception_;
};
class FailureOrVoid {
 public:
  FailureOrVoid() {}
  explicit FailureOrVoid(Exception* e) : exception_(e) {}
  FailureOrVoid(FailureOrVoid&& o) = default;
  bool is_failure() const { return exception_ != nullptr; }
  bool is_success() const { return !is_failure(); }
  const Exception& exception() const { return *exception_; }
  Exception* release_exception() { return exception_.release(); }
 private:
  std::unique_ptr<Exception> exception_;
};

// SucceedOrDie (utils/exception/failureor.h:442-466): "turn exceptions into runtime crashes" -- the adapters user code wraps
// around CreateCursor() / Bind() when a failure is a programming error (test/guide/primer.cc:286-291).
namespace internal {
inline void Die(const Exception& e) { fprintf(stderr, "SucceedOrDie: %s\n", e.PrintStackTrace().c_str()); abort(); }
}  // namespace internal
inline void SucceedOrDie(FailureOrVoid result) { if (result.is_failure()) internal::Die(result.exception()); }
template <typename T> T SucceedOrDie(FailureOr<T> result) { if (result.is_failure()) internal::Die(result.exception()); return result.get(); }
template <typename T> T* SucceedOrDie(FailureOrOwned<T> result) { if (result.is_failure()) internal::Die(result.exception()); return result.release(); }   // ownership passes to the caller

// ---- schema (base/infrastructure/tuple_schema.h:77,126) ---------------------------------
class Attribute {
 public:
  Attribute(const std::string& name, DataType type, Nullability nullability) : name_(name), type_(type), nullability_(nullability) {}
  const std::string& name() const { return name_; }
  DataType type() const { return type_; }
  Nullability nullability() const { return nullability_; }
  bool is_nullable() const { return nullability_ == NULLABLE; }
 private:
  std::string name_;
  DataType type_;
  Nullability nullability_;
};

class TupleSchema {
 public:
  TupleSchema() {}
  static TupleSchema Singleton(const std::string& name, DataType type, Nullability n) { TupleSchema s; s.add_attribute(Attribute(name, type, n)); return s; }
  bool add_attribute(const Attribute& a) { if (LookupAttributePosition(a.name()) >= 0) return false; attrs_.push_back(a); return true; }
  int attribute_count() const { return static_cast<int>(attrs_.size()); }
  const Attribute& attribute(int i) const { return attrs_[i]; }
  int LookupAttributePosition(const std::string& name) const {
    for (size_t i = 0; i < attrs_.size(); ++i) if (attrs_[i].name() == name) return static_cast<int>(i);
    return -1;
  }
 private:
  std::vector<Attribute> attrs_;
};

inline size_t SizeOfDataType(DataType t) {
  switch (t) { case INT32: case UINT32: case FLOAT: case DATE: return 4; case INT64: case UINT64: case DOUBLE: case DATETIME: return 8; case BOOL: return 1;
               case STRING: return sizeof(StringPiece); default: return 0; }
}

// ---- memory (base/memory/memory.h:100-233, 240, 465-520) over ssgpu_allocator_* -------------------------------
class BufferAllocator;
// Owns one block of a BufferAllocator; destroying it returns the block (memory.h:55-98).
class Buffer {
 public:
  ~Buffer();
  void* data() const { return data_; }
  size_t size() const { return size_; }
 private:
  friend class BufferAllocator;
  Buffer(void* data, size_t size, BufferAllocator* a) : data_(data), size_(size), allocator_(a) {}
  void* data_; size_t size_; BufferAllocator* allocator_;
};

// ---- TypeTraits / variant pointers (base/infrastructure/types.h:300-420, variant_pointer.h:87-139, bit_pointers.h) -------
template <DataType type> struct TypeTraits;
template <> struct TypeTraits<INT32> { typedef int32 cpp_type; };
template <> struct TypeTraits<INT64> { typedef int64 cpp_type; };
template <> struct TypeTraits<UINT32> { typedef uint32 cpp_type; };
template <> struct TypeTraits<UINT64> { typedef uint64 cpp_type; };
template <> struct TypeTraits<FLOAT> { typedef float cpp_type; };
template <> struct TypeTraits<DOUBLE> { typedef double cpp_type; };
template <> struct TypeTraits<BOOL> { typedef bool cpp_type; };
template <> struct TypeTraits<DATE> { typedef int32 cpp_type; };
template <> struct TypeTraits<DATETIME> { typedef int64 cpp_type; };
template <> struct TypeTraits<STRING> { typedef StringPiece cpp_type; };
template <> struct TypeTraits<BINARY> { typedef StringPiece cpp_type; };

typedef const bool* bool_const_ptr;   // one byte per row (bit_pointers.h:529-533, the non-bit-packed build)
typedef bool* bool_ptr;

class VariantConstPointer {
 public:
  VariantConstPointer() : pointer_(nullptr) {}
  VariantConstPointer(const void* pointer) : pointer_(pointer) {}   // NOLINT(runtime/explicit): as the reference
  template <DataType type> const typename TypeTraits<type>::cpp_type* as() const { return static_cast<const typename TypeTraits<type>::cpp_type*>(pointer_); }
  const StringPiece* as_variable_length() const { return static_cast<const StringPiece*>(pointer_); }
  const void* raw() const { return pointer_; }
  bool is_null() const { return pointer_ == nullptr; }
  VariantConstPointer offset(int64 rows, DataType type) const { return VariantConstPointer(static_cast<const char*>(pointer_) + rows * static_cast<int64>(SizeOfDataType(type))); }
 private:
  friend bool operator==(VariantConstPointer a, VariantConstPointer b);
  const void* pointer_;
};
inline bool operator==(VariantConstPointer a, VariantConstPointer b) { return a.pointer_ == b.pointer_; }

// ---- View (base/infrastructure/block.h:55-402): N (data, is_null) pairs + row count -----
class Column {
 public:
  const Attribute& attribute() const { return *attribute_; }
  VariantConstPointer data() const { return data_; }
  VariantConstPointer data_plus_offset(rowid_t offset) const { return data_.offset(offset, attribute_->type()); }
  bool_const_ptr is_null() const { return is_null_; }   // nullptr => no NULLs
  bool_const_ptr is_null_plus_offset(rowcount_t offset) const { return is_null_ ? is_null_ + offset : nullptr; }
  // typed_data<INT32>() as in the reference (block.h:90-96) ...
  template <DataType type> const typename TypeTraits<type>::cpp_type* typed_data() const { return data_.as<type>(); }
  // ... and by C++ type, for generic code on this side
  template <typename T> const T* typed_data() const { return static_cast<const T*>(data_.raw()); }
  const StringPiece* variable_length_data() const { return data_.as_variable_length(); }
  void Reset(VariantConstPointer data, bool_const_ptr is_null) { data_ = data; is_null_ = is_null; }
  void ResetFrom(const Column& other) { Reset(other.data(), other.is_null()); }
  void ResetFromPlusOffset(const Column& other, rowcount_t offset) { Reset(other.data_plus_offset(static_cast<rowid_t>(offset)), other.is_null_plus_offset(offset)); }
  void ResetIsNull(bool_const_ptr is_null) { if (attribute_->is_nullable()) is_null_ = is_null; }
 private:
  friend class View;
  Column() : attribute_(nullptr), is_null_(nullptr) {}
  const Attribute* attribute_;   // points into the owning View's schema
  VariantConstPointer data_;
  bool_const_ptr is_null_;
};

class View {
 public:
  explicit View(const TupleSchema& schema) : schema_(schema), columns_(new Column[Count(schema)]), row_count_(0) { Init(); }
  View(const View& other) : schema_(other.schema()), columns_(new Column[Count(other.schema())]), row_count_(0) { Init(); ResetFrom(other); }
  View(const View& other, rowcount_t offset, rowcount_t row_count) : schema_(other.schema()), columns_(new Column[Count(other.schema())]), row_count_(0) {
    Init(); ResetFromSubRange(other, offset, row_count);
  }
  View(const Column& column, rowcount_t row_count)
      : schema_(TupleSchema::Singleton(column.attribute().name(), column.attribute().type(), column.attribute().nullability())), columns_(new Column[1]), row_count_(row_count) {
    Init(); mutable_column(0)->ResetFrom(column);
  }
  const TupleSchema& schema() const { return schema_; }
  int column_count() const { return schema_.attribute_count(); }
  const Column& column(int i) const { return columns_[i]; }
  Column* mutable_column(int i) { return &columns_[i]; }
  rowcount_t row_count() const { return row_count_; }
  void set_row_count(rowcount_t n) { row_count_ = n; }
  void ResetFrom(const View& other) {
    for (int i = 0; i < column_count(); ++i) mutable_column(i)->ResetFrom(other.column(i));
    set_row_count(other.row_count());
  }
  void ResetFromSubRange(const View& other, rowcount_t offset, rowcount_t row_count) {
    for (int i = 0; i < column_count(); ++i) mutable_column(i)->ResetFromPlusOffset(other.column(i), offset);
    set_row_count(row_count);
  }
  void Advance(rowcount_t offset) {
    for (int i = 0; i < column_count(); ++i) { Column* c = mutable_column(i); c->ResetFromPlusOffset(*c, offset); }
    row_count_ -= (row_count_ < offset ? row_count_ : offset);
  }
 private:
  static size_t Count(const TupleSchema& s) { return static_cast<size_t>(s.attribute_count() > 0 ? s.attribute_count() : 1); }
  void Init() { for (int i = 0; i < schema_.attribute_count(); ++i) columns_[i].attribute_ = &schema_.attribute(i); }
  View& operator=(const View& other);   // not assignable, as the reference's (block.h:396)
  const TupleSchema schema_;
  std::unique_ptr<Column[]> columns_;
  rowcount_t row_count_;
};

// Columns that already live in device memory (e.g. the unpacked partial tables of a sharded aggregate, another plan's
// result): scanned with ScanDeviceView, never copied.  The caller keeps the memory alive while cursors use it.
struct DeviceView {
  TupleSchema schema;
  std::vector<ssgpu_column> columns;   // one (data, is_null) pair of DEVICE pointers per attribute
  rowcount_t row_count = 0;
};

// ---- expressions (expression/base/expression.h, core/*_expressions.h) --------------------
class BoundExpressionTree;
class BufferAllocator;
class Expression {
 public:
  Expression(int kind, int op, int dtype, int64_t i64, double f64, const std::string& name) : kind(kind), op(op), dtype(dtype), i64(i64), f64(f64), name(name) {}
  virtual ~Expression() {}
  int kind, op, dtype;
  int64_t i64;
  double f64;
  std::string name;
  std::string sval;          // payload of a ConstString
  std::vector<std::unique_ptr<const Expression>> args;
  // Expression::Bind(input_schema, allocator, max_row_count) (expression.h:158-160); defined below
  FailureOrOwned<BoundExpressionTree> Bind(const TupleSchema& input_schema, BufferAllocator* allocator, rowcount_t max_row_count) const;
};
namespace internal {
inline Expression* Node(int kind, int op = 0, int dtype = 0, int64_t i64 = 0, double f64 = 0, const std::string& name = "") { return new Expression(kind, op, dtype, i64, f64, name); }
inline const Expression* Op(int op, const Expression* a, const Expression* b = nullptr, const Expression* c = nullptr) {
  Expression* e = Node(SSGPU_EXPR_OP, op);
  e->args.emplace_back(a); if (b) e->args.emplace_back(b); if (c) e->args.emplace_back(c);
  return e;
}
}  // namespace internal
inline const Expression* NamedAttribute(const std::string& name) { return internal::Node(SSGPU_EXPR_ATTR_NAMED, 0, 0, 0, 0, name); }
inline const Expression* AttributeAt(size_t position) { return internal::Node(SSGPU_EXPR_ATTR_AT, 0, 0, static_cast<int64_t>(position)); }
inline const Expression* ConstInt32(int32_t v) { return internal::Node(SSGPU_EXPR_CONST, 0, INT32, v); }
inline const Expression* ConstInt64(int64_t v) { return internal::Node(SSGPU_EXPR_CONST, 0, INT64, v); }
inline const Expression* ConstUint32(uint32_t v) { return internal::Node(SSGPU_EXPR_CONST, 0, UINT32, v); }
inline const Expression* ConstUint64(uint64_t v) { return internal::Node(SSGPU_EXPR_CONST, 0, UINT64, static_cast<int64_t>(v)); }
inline const Expression* ConstFloat(float v) { return internal::Node(SSGPU_EXPR_CONST, 0, FLOAT, 0, v); }
inline const Expression* ConstDouble(double v) { return internal::Node(SSGPU_EXPR_CONST, 0, DOUBLE, 0, v); }
inline const Expression* ConstString(const StringPiece& v) { Expression* e = internal::Node(SSGPU_EXPR_CONST, 0, STRING); e->sval = v.ToString(); return e; }
inline const Expression* ConstBool(bool v) { return internal::Node(SSGPU_EXPR_CONST, 0, BOOL, v ? 1 : 0); }
inline const Expression* Null(DataType t) { return internal::Node(SSGPU_EXPR_NULL, 0, t); }
inline const Expression* Plus(const Expression* a, const Expression* b) { return internal::Op(0, a, b); }
inline const Expression* Multiply(const Expression* a, const Expression* b) { return internal::Op(4, a, b); }
inline const Expression* Minus(const Expression* a, const Expression* b) { return internal::Op(8, a, b); }
inline const Expression* DivideQuiet(const Expression* a, const Expression* b) { return internal::Op(13, a, b); }
inline const Expression* DivideNulling(const Expression* a, const Expression* b) { return internal::Op(14, a, b); }
inline const Expression* DivideSignaling(const Expression* a, const Expression* b) { return internal::Op(15, a, b); }
inline const Expression* Divide(const Expression* a, const Expression* b) { return DivideSignaling(a, b); }
inline const Expression* CppDivideNulling(const Expression* a, const Expression* b) { return internal::Op(18, a, b); }
inline const Expression* CppDivideSignaling(const Expression* a, const Expression* b) { return internal::Op(19, a, b); }
inline const Expression* CppDivide(const Expression* a, const Expression* b) { return CppDivideSignaling(a, b); }
inline const Expression* ModulusNulling(const Expression* a, const Expression* b) { return internal::Op(26, a, b); }
inline const Expression* ModulusSignaling(const Expression* a, const Expression* b) { return internal::Op(27, a, b); }
inline const Expression* Modulus(const Expression* a, const Expression* b) { return ModulusSignaling(a, b); }
inline const Expression* Negate(const Expression* a) { return internal::Op(36, a); }
inline const Expression* And(const Expression* a, const Expression* b) { return internal::Op(40, a, b); }
inline const Expression* Or(const Expression* a, const Expression* b) { return internal::Op(44, a, b); }
inline const Expression* AndNot(const Expression* a, const Expression* b) { return internal::Op(48, a, b); }
inline const Expression* Not(const Expression* a) { return internal::Op(52, a); }
inline const Expression* Xor(const Expression* a, const Expression* b) { return internal::Op(56, a, b); }
inline const Expression* BitwiseAnd(const Expression* a, const Expression* b) { return internal::Op(60, a, b); }
inline const Expression* BitwiseOr(const Expression* a, const Expression* b) { return internal::Op(64, a, b); }
inline const Expression* BitwiseNot(const Expression* a) { return internal::Op(68, a); }
inline const Expression* BitwiseXor(const Expression* a, const Expression* b) { return internal::Op(72, a, b); }
inline const Expression* ShiftLeft(const Expression* a, const Expression* b) { return internal::Op(76, a, b); }
inline const Expression* ShiftRight(const Expression* a, const Expression* b) { return internal::Op(80, a, b); }
inline const Expression* BitwiseAndNot(const Expression* a, const Expression* b) { return internal::Op(84, a, b); }
inline const Expression* Equal(const Expression* a, const Expression* b) { return internal::Op(100, a, b); }
inline const Expression* NotEqual(const Expression* a, const Expression* b) { return internal::Op(104, a, b); }
inline const Expression* Less(const Expression* a, const Expression* b) { return internal::Op(116, a, b); }
inline const Expression* LessOrEqual(const Expression* a, const Expression* b) { return internal::Op(120, a, b); }
inline const Expression* Greater(const Expression* a, const Expression* b) { return internal::Op(SSGPU_OP_GREATER, a, b); }
inline const Expression* GreaterOrEqual(const Expression* a, const Expression* b) { return internal::Op(SSGPU_OP_GREATER_OR_EQUAL, a, b); }
// exact math family (expression/core/math_expressions.h:78-126; IsOdd/IsEven: comparison_expressions.h)
inline const Expression* Abs(const Expression* a) { return internal::Op(360, a); }
inline const Expression* Round(const Expression* a) { return internal::Op(300, a); }
inline const Expression* Ceil(const Expression* a) { return internal::Op(342, a); }
inline const Expression* Floor(const Expression* a) { return internal::Op(346, a); }
inline const Expression* Trunc(const Expression* a) { return internal::Op(304, a); }
inline const Expression* RoundToInt(const Expression* a) { return internal::Op(316, a); }
inline const Expression* CeilToInt(const Expression* a) { return internal::Op(308, a); }
inline const Expression* FloorToInt(const Expression* a) { return internal::Op(312, a); }
inline const Expression* SqrtQuiet(const Expression* a) { return internal::Op(333, a); }
inline const Expression* SqrtNulling(const Expression* a) { return internal::Op(334, a); }
inline const Expression* SqrtSignaling(const Expression* a) { return internal::Op(335, a); }
// libm family (expression/core/math_expressions.h): within a few ULP of the host libm
inline const Expression* RoundWithPrecision(const Expression* a, const Expression* precision) { return internal::Op(SSGPU_OP_ROUND_WITH_PRECISION, a, precision); }
inline const Expression* Exp(const Expression* a) { return internal::Op(320, a); }
inline const Expression* LnQuiet(const Expression* a) { return internal::Op(325, a); }
inline const Expression* LnNulling(const Expression* a) { return internal::Op(326, a); }
inline const Expression* Log10Quiet(const Expression* a) { return internal::Op(329, a); }
inline const Expression* Log10Nulling(const Expression* a) { return internal::Op(330, a); }
inline const Expression* Log2Quiet(const Expression* a) { return internal::Op(357, a); }
inline const Expression* Log2Nulling(const Expression* a) { return internal::Op(358, a); }
inline const Expression* PowerQuiet(const Expression* a, const Expression* b) { return internal::Op(353, a, b); }
inline const Expression* PowerNulling(const Expression* a, const Expression* b) { return internal::Op(354, a, b); }
inline const Expression* PowerSignaling(const Expression* a, const Expression* b) { return internal::Op(355, a, b); }
inline const Expression* Sin(const Expression* a) { return internal::Op(800, a); }
inline const Expression* Cos(const Expression* a) { return internal::Op(804, a); }
inline const Expression* Tan(const Expression* a) { return internal::Op(808, a); }
inline const Expression* Asin(const Expression* a) { return internal::Op(812, a); }
inline const Expression* Acos(const Expression* a) { return internal::Op(816, a); }
inline const Expression* Atan(const Expression* a) { return internal::Op(820, a); }
inline const Expression* Atan2(const Expression* x, const Expression* y) { return internal::Op(824, x, y); }
inline const Expression* Sinh(const Expression* a) { return internal::Op(828, a); }
inline const Expression* Cosh(const Expression* a) { return internal::Op(832, a); }
inline const Expression* Tanh(const Expression* a) { return internal::Op(836, a); }
inline const Expression* Asinh(const Expression* a) { return internal::Op(840, a); }
inline const Expression* Acosh(const Expression* a) { return internal::Op(844, a); }
inline const Expression* Atanh(const Expression* a) { return internal::Op(848, a); }
inline const Expression* IsFinite(const Expression* a) { return internal::Op(148, a); }
inline const Expression* IsInf(const Expression* a) { return internal::Op(152, a); }
inline const Expression* IsNaN(const Expression* a) { return internal::Op(156, a); }
inline const Expression* IsNormal(const Expression* a) { return internal::Op(160, a); }
inline const Expression* IsOdd(const Expression* a) { return internal::Op(140, a); }
inline const Expression* IsEven(const Expression* a) { return internal::Op(144, a); }
// owning list of expressions (expression/base/expression.h): the arguments of Case / In
class ExpressionList {
 public:
  ExpressionList* add(const Expression* e) { items.emplace_back(e); return this; }
  std::vector<std::unique_ptr<const Expression>> items;
};
// CASE arg0 WHEN arg2 THEN arg3 [...] ELSE arg1 (elementary_expressions.h:91-93); takes ownership
inline const Expression* Case(ExpressionList* arguments) {
  std::unique_ptr<ExpressionList> own(arguments);
  Expression* e = internal::Node(SSGPU_EXPR_OP, 200);
  for (auto& a : own->items) e->args.emplace_back(a.release());
  return e;
}
// needle IN (haystack...) with SQL NULL semantics (comparison_expressions.h:75-89); takes ownership
inline const Expression* In(const Expression* needle, ExpressionList* haystack) {
  std::unique_ptr<ExpressionList> own(haystack);
  Expression* e = internal::Node(SSGPU_EXPR_OP, 208);
  e->args.emplace_back(needle);
  for (auto& a : own->items) e->args.emplace_back(a.release());
  return e;
}
inline const Expression* If(const Expression* c, const Expression* t, const Expression* e) { return internal::Op(204, c, t, e); }
inline const Expression* NullingIf(const Expression* c, const Expression* t, const Expression* e) { return internal::Op(SSGPU_OP_NULLING_IF, c, t, e); }
inline const Expression* IfNull(const Expression* a, const Expression* b) { return internal::Op(220, a, b); }
inline const Expression* IsNull(const Expression* a) { return internal::Op(224, a); }
inline const Expression* CastTo(DataType t, const Expression* a) { Expression* e = internal::Node(SSGPU_EXPR_CAST, 0, t); e->args.emplace_back(a); return e; }
inline const Expression* Alias(const std::string& new_name, const Expression* a) { Expression* e = internal::Node(SSGPU_EXPR_ALIAS, 0, 0, 0, 0, new_name); e->args.emplace_back(a); return e; }

class CompoundExpression : public Expression {
 public:
  CompoundExpression() : Expression(SSGPU_EXPR_COMPOUND, 0, 0, 0, 0, "") {}
  CompoundExpression* Add(const Expression* argument) { args.emplace_back(argument); return this; }
  CompoundExpression* AddAs(const std::string& alias, const Expression* argument) { args.emplace_back(Alias(alias, argument)); return this; }
};

// ---- projectors (base/infrastructure/projector.h) ------------------------------------------
class SingleSourceProjector {
 public:
  virtual ~SingleSourceProjector() {}
  struct Entry { int kind; int position; std::string name, alias; int source = 0; };
  std::vector<Entry> entries;
};
inline const SingleSourceProjector* ProjectAllAttributes(const std::string& prefix = "") { auto* p = new SingleSourceProjector; p->entries.push_back({SSGPU_PROJ_ALL, 0, "", prefix}); return p; }
inline const SingleSourceProjector* ProjectNamedAttribute(const std::string& name) { auto* p = new SingleSourceProjector; p->entries.push_back({SSGPU_PROJ_NAMED, 0, name, ""}); return p; }
inline const SingleSourceProjector* ProjectNamedAttributeAs(const std::string& name, const std::string& alias) { auto* p = new SingleSourceProjector; p->entries.push_back({SSGPU_PROJ_NAMED_AS, 0, name, alias}); return p; }
inline const SingleSourceProjector* ProjectAttributeAt(int position) { auto* p = new SingleSourceProjector; p->entries.push_back({SSGPU_PROJ_AT, position, "", ""}); return p; }
inline const SingleSourceProjector* ProjectNamedAttributes(const std::vector<std::string>& names) { auto* p = new SingleSourceProjector; for (auto& n : names) p->entries.push_back({SSGPU_PROJ_NAMED, 0, n, ""}); return p; }
class CompoundSingleSourceProjector : public SingleSourceProjector {
 public:
  CompoundSingleSourceProjector* add(const SingleSourceProjector* p) { std::unique_ptr<const SingleSourceProjector> own(p); for (auto& e : p->entries) entries.push_back(e); return this; }
};

// (source index, projector) pairs over the inputs of a join (base/infrastructure/projector.h:405-441)
class MultiSourceProjector {
 public:
  virtual ~MultiSourceProjector() {}
  std::vector<SingleSourceProjector::Entry> entries;
};
class CompoundMultiSourceProjector : public MultiSourceProjector {
 public:
  CompoundMultiSourceProjector* add(int source_index, const SingleSourceProjector* p) {
    std::unique_ptr<const SingleSourceProjector> own(p);
    for (auto e : p->entries) { e.source = source_index; entries.push_back(e); }
    return this;
  }
};

// ---- specifications (cursor/core/aggregate.h:28-205, infrastructure/ordering.h:48-101) -----
class AggregationSpecification {
 public:
  struct Element { Aggregation aggregation; std::string input, output; int output_type; bool distinct; };
  AggregationSpecification* AddAggregation(Aggregation a, const std::string& in, const std::string& out) { elements.push_back({a, in, out, -1, false}); return this; }
  AggregationSpecification* AddDistinctAggregation(Aggregation a, const std::string& in, const std::string& out) { elements.push_back({a, in, out, -1, true}); return this; }
  AggregationSpecification* AddAggregationWithDefinedOutputType(Aggregation a, const std::string& in, const std::string& out, DataType t) { elements.push_back({a, in, out, t, false}); return this; }
  std::vector<Element> elements;
};
// cursor/core/aggregate.h:160-205.  max_unique_keys_in_result: the result keeps the first (limit + 1) distinct keys in
// first-seen order and every row with another key is aggregated into the last of those rows (row_hash_set.cc:500-511);
// default kint64max = no limit.  Memory quota / estimated row count: no device counterpart (tables are sized by run feedback).
class GroupAggregateOptions {
 public:
  GroupAggregateOptions() : max_unique_keys_in_result_(INT64_MAX), memory_quota_(0) {}
  int64_t max_unique_keys_in_result() const { return max_unique_keys_in_result_; }
  GroupAggregateOptions* set_max_unique_keys_in_result_(int64_t n) { max_unique_keys_in_result_ = n; return this; }
  // aggregate.h:170-175.  BestEffortGroupAggregate: the result block of a view holds quota / (bytes of a result row) groups
  // (ssgpu.h ssgpu_plan_run_best_effort); GroupAggregate tables are sized by run feedback and ignore it.  0 / never set = no quota.
  GroupAggregateOptions* set_memory_quota(size_t bytes) { memory_quota_ = bytes > (size_t(1) << 62) ? 0 : static_cast<int64_t>(bytes); return this; }
  int64_t memory_quota() const { return memory_quota_; }
  GroupAggregateOptions* set_enforce_quota(bool) { return this; }
  GroupAggregateOptions* set_estimated_result_row_count(int64_t) { return this; }
  // ssgpu_op.option0: 0 = no limit, n > 0 = limit n, -1 = limit 0
  int64_t option0() const { return max_unique_keys_in_result_ == INT64_MAX ? 0 : max_unique_keys_in_result_ == 0 ? -1 : max_unique_keys_in_result_; }
 private:
  int64_t max_unique_keys_in_result_;
  int64_t memory_quota_;
};
class SortOrder {
 public:
  SortOrder* add(const SingleSourceProjector* projector, ColumnOrder order) {
    std::unique_ptr<const SingleSourceProjector> own(projector);
    for (auto& e : projector->entries) keys.push_back(std::make_pair(e.name, order));
    return this;
  }
  std::vector<std::pair<std::string, ColumnOrder>> keys;
};

// ---- process-wide device context -------------------------------------------------------------
namespace internal {
struct Context {
  ssgpu_ctx* ctx = nullptr;
  // > 0: a plan with a chunked form (ssgpu.h "CHUNKED STAGING": ScalarAggregate / row-local / GroupAggregate-first) over a host View of more rows than this is staged in chunks of this many rows (ssgpu_plan_run_host: the
  // copy of chunk k + 1 overlaps the kernel over chunk k; inputs larger than device memory run) -- SetHostStagingChunkRows below
  rowcount_t host_chunk_rows = 0;
  Context() { if (ssgpu_ctx_create(0, &ctx) != SSGPU_OK) ssgpu_ctx_create(-1, &ctx); }  // bind-only without a GPU
  ~Context() { ssgpu_ctx_destroy(ctx); }
  static Context& Get() { static Context c; return c; }
};
}  // namespace internal

// Extension of this library (no counterpart in the reference, whose cursors stream by construction): ScalarAggregates over host
// Views of more than `rows` rows are staged through the device in chunks of `rows` rows instead of being uploaded whole
// (0 = off, the default).  Takes effect for cursors created afterwards.
inline void SetHostStagingChunkRows(rowcount_t rows) { internal::Context::Get().host_chunk_rows = rows; }

// Pinned-host allocator with an optional soft quota.  Allocate / BestEffortAllocate return NULL when the quota (or the
// host) cannot serve `minimal` -- callers turn that into ERROR_MEMORY_EXCEEDED, as the reference's do.
class BufferAllocator {
 public:
  virtual ~BufferAllocator() { if (a_) ssgpu_allocator_destroy(a_); }
  Buffer* Allocate(size_t requested) { return BestEffortAllocate(requested, requested); }
  Buffer* BestEffortAllocate(size_t requested, size_t minimal) {
    void* p = nullptr; size_t granted = 0;
    if (ssgpu_allocator_allocate(a_, requested, minimal, &p, &granted) != SSGPU_OK) return nullptr;
    return new Buffer(p, granted, this);
  }
  // On failure returns false and leaves the buffer as it was (memory.h:140-160).
  bool Reallocate(size_t requested, Buffer* buffer) { return BestEffortReallocate(requested, requested, buffer); }
  bool BestEffortReallocate(size_t requested, size_t minimal, Buffer* buffer) {
    void* p = nullptr; size_t granted = 0;
    if (ssgpu_allocator_reallocate(a_, buffer->data_, requested, minimal, &p, &granted) != SSGPU_OK) return false;
    buffer->data_ = p; buffer->size_ = granted;
    return true;
  }
  size_t Available() const { const int64_t v = ssgpu_allocator_available(a_); return v < 0 ? 0 : static_cast<size_t>(v); }
  size_t GetUsage() const { return static_cast<size_t>(ssgpu_allocator_allocated(a_)); }
  bool has_quota() const { return quota_ >= 0; }
 protected:
  explicit BufferAllocator(int64_t quota) : quota_(quota) { ssgpu_allocator_create(internal::Context::Get().ctx, quota, &a_); }
 private:
  friend class Buffer;
  ssgpu_allocator* a_ = nullptr;
  int64_t quota_;
};
inline Buffer::~Buffer() { ssgpu_allocator_free(allocator_->a_, data_); }

class HeapBufferAllocator : public BufferAllocator {
 public:
  static HeapBufferAllocator* Get() { static HeapBufferAllocator a; return &a; }   // memory.h:240-250
 private:
  HeapBufferAllocator() : BufferAllocator(-1) {}
};
// MemoryLimit(quota): a soft quota; the delegate of the reference's constructor is always the pinned-host heap here.
class MemoryLimit : public BufferAllocator {
 public:
  explicit MemoryLimit(size_t quota) : BufferAllocator(static_cast<int64_t>(quota)) {}
  MemoryLimit(size_t quota, bool /*enforced*/, BufferAllocator* /*delegate*/) : BufferAllocator(static_cast<int64_t>(quota)) {}
};

namespace internal {
// One order-preserving dictionary per cursor / evaluated View (ssgpu_dict_*): STRING cells <-> INT32 codes.
struct Dictionary {
  ssgpu_dict* d = nullptr;
  ~Dictionary() { if (d) ssgpu_dict_destroy(d); }
  int Build(const std::vector<StringPiece>& values) {
    if (d) { ssgpu_dict_destroy(d); d = nullptr; }
    std::vector<const char*> ptr; std::vector<int32_t> len;
    for (auto& v : values) { ptr.push_back(v.data()); len.push_back(static_cast<int32_t>(v.size())); }
    return ssgpu_dict_create(ptr.data(), len.data(), static_cast<int64_t>(values.size()), &d);
  }
  int Encode(const StringPiece* cells, const bool* is_null, rowcount_t n, std::vector<int32_t>* codes) const {
    std::vector<const char*> ptr(static_cast<size_t>(n)); std::vector<int32_t> len(static_cast<size_t>(n));
    for (rowcount_t i = 0; i < n; ++i) { ptr[i] = cells[i].data(); len[i] = static_cast<int32_t>(cells[i].size()); }
    codes->assign(static_cast<size_t>(std::max<rowcount_t>(n, 1)), 0);
    return ssgpu_dict_encode(d, ptr.data(), len.data(), reinterpret_cast<const uint8_t*>(is_null), static_cast<int64_t>(n), codes->data());
  }
  StringPiece Decode(int32_t code) const {
    const char* b = nullptr; int32_t n = 0;
    return ssgpu_dict_decode(d, code, &b, &n) == SSGPU_OK ? StringPiece(b, static_cast<size_t>(n)) : StringPiece();
  }
  static void Collect(const View& v, std::vector<StringPiece>* out) {
    for (int i = 0; i < v.schema().attribute_count(); ++i) {
      if (v.schema().attribute(i).type() != STRING) continue;
      const StringPiece* cells = v.column(i).typed_data<StringPiece>();
      const bool* z = v.column(i).is_null();
      for (rowcount_t r = 0; r < v.row_count(); ++r) if (!z || !z[r]) out->push_back(cells[r]);
    }
  }
  static bool HasStrings(const TupleSchema& s) {
    for (int i = 0; i < s.attribute_count(); ++i) if (s.attribute(i).type() == STRING) return true;
    return false;
  }
};
// host View -> device block (STRING columns as dictionary codes)
inline int UploadView(ssgpu_ctx* ctx, const View* v, const Dictionary* dict, ssgpu_block** out) {
  const TupleSchema& s = v->schema();
  std::vector<ssgpu_attr> attrs;
  for (int i = 0; i < s.attribute_count(); ++i) attrs.push_back({s.attribute(i).name().c_str(), s.attribute(i).type(), s.attribute(i).nullability()});
  int rc = ssgpu_block_create(ctx, attrs.data(), static_cast<int32_t>(attrs.size()), static_cast<int64_t>(std::max<rowcount_t>(v->row_count(), 1)), out);
  std::vector<int32_t> codes;
  for (int i = 0; rc == SSGPU_OK && i < s.attribute_count() && v->row_count() > 0; ++i) {
    const void* data = v->column(i).data().raw();
    if (s.attribute(i).type() == STRING) {
      rc = dict && dict->d ? dict->Encode(v->column(i).typed_data<StringPiece>(), v->column(i).is_null(), v->row_count(), &codes) : SSGPU_ERROR_INVALID_ARGUMENT_VALUE;
      data = codes.data();
      if (rc != SSGPU_OK) break;
    }
    rc = ssgpu_block_upload(*out, i, data, reinterpret_cast<const uint8_t*>(v->column(i).is_null()), 0, static_cast<int64_t>(v->row_count()));
    if (rc == SSGPU_OK && s.attribute(i).type() == STRING) rc = ssgpu_ctx_synchronize(ctx);   // `codes` is reused by the next column
  }
  if (rc == SSGPU_OK) rc = ssgpu_block_set_row_count(*out, static_cast<int64_t>(v->row_count()));
  return rc;
}
// finished result -> host column pointers (STRING columns decoded into `cells`)
inline int FetchResult(ssgpu_result* res, const TupleSchema& schema, const Dictionary* dict, rowcount_t* total,
                       std::vector<const void*>* data, std::vector<const uint8_t*>* nulls, std::vector<std::vector<StringPiece>>* cells) {
  const int64_t fetched_rows = ssgpu_result_row_count(res);
  if (fetched_rows < 0) return SSGPU_ERROR_HIP;
  *total = static_cast<rowcount_t>(fetched_rows);
  data->clear(); nulls->clear(); cells->assign(static_cast<size_t>(schema.attribute_count()), std::vector<StringPiece>());
  for (int i = 0; i < schema.attribute_count(); ++i) {
    const void* d = nullptr; const uint8_t* z = nullptr;
    const int rc = ssgpu_result_column(res, i, &d, &z);
    if (rc != SSGPU_OK) return rc;
    if (schema.attribute(i).type() == STRING) {
      std::vector<StringPiece>& c = (*cells)[static_cast<size_t>(i)];
      c.resize(static_cast<size_t>(std::max<rowcount_t>(*total, 1)));
      // a CONCAT column's cells are codes of a dictionary the RESULT owns (its strings exist nowhere else); every other STRING
      // column holds codes of the cursor's dictionary
      if (const ssgpu_dict* own = ssgpu_result_column_dict(res, i)) {
        for (rowcount_t r = 0; r < *total; ++r) {
          const char* b = nullptr; int32_t n = 0;
          c[r] = (z && z[r]) || ssgpu_dict_decode(own, static_cast<const int32_t*>(d)[r], &b, &n) != SSGPU_OK ? StringPiece() : StringPiece(b, static_cast<size_t>(n));
        }
      } else {
        for (rowcount_t r = 0; r < *total; ++r) c[r] = (z && z[r]) || !dict ? StringPiece() : dict->Decode(static_cast<const int32_t*>(d)[r]);
      }
      d = c.data();
    }
    data->push_back(d); nulls->push_back(z);
  }
  return SSGPU_OK;
}
}  // namespace internal

// ---- cursors (cursor/base/cursor.h:42-226) -----------------------------------------------------
class ResultView {
 public:
  static ResultView Success(const View* v) { ResultView r; r.view_ = v; return r; }
  static ResultView EOS() { ResultView r; r.status_ = kEos; return r; }
  static ResultView BOS() { ResultView r; r.status_ = kBos; return r; }
  static ResultView WaitingOnBarrier() { ResultView r; r.status_ = kBarrier; return r; }
  static ResultView Failure(Exception* e) { ResultView r; r.exception_.reset(e); return r; }
  bool has_data() const { return view_ != nullptr; }
  bool is_done() const { return is_eos() || is_failure(); }
  bool is_eos() const { return status_ == kEos; }
  bool is_bos() const { return status_ == kBos; }
  bool is_waiting_on_barrier() const { return status_ == kBarrier; }
  bool is_failure() const { return exception_ != nullptr; }
  const View& view() const { return *view_; }
  const Exception& exception() const { return *exception_; }
  // ownership of the exception passes to the caller (a copy: the ResultView itself stays copyable, cursor.h:120)
  Exception* release_exception() { Exception* e = exception_ ? new Exception(*exception_) : nullptr; exception_.reset(); return e; }
 private:
  enum Status { kData, kEos, kBos, kBarrier };
  ResultView() : view_(nullptr), status_(kData) {}
  const View* view_;
  Status status_;
  std::shared_ptr<Exception> exception_;
};
inline const View& SucceedOrDie(ResultView result_view) {   // cursor.h:124-127
  if (result_view.is_failure()) internal::Die(result_view.exception());
  return result_view.view();
}

// cursor/proto/cursors.proto:13-67 (the ids of the cursors this path has)
enum CursorId { FILE_INPUT = 3, VIEW = 7, AGGREGATE_CLUSTERS = 8, BEST_EFFORT_GROUP_AGGREGATE = 10, COMPUTE = 15, FILTER = 16, GROUP_AGGREGATE = 18, HASH_JOIN = 19, PROJECT = 26,
                SCALAR_AGGREGATE = 29, SORT = 30, UNKNOWN_ID = 42 };
class CursorTransformer;

class Operation;

// The reference's pure interface (cursor/base/cursor.h:131-226): user code holds Cursor*, calls schema() / Next() /
// Interrupt(), and may implement its own cursors against it.
class Cursor {
 public:
  static const rowcount_t kDefaultRowCount = 1024;  // cursor.h:133
  virtual ~Cursor() {}
  virtual const TupleSchema& schema() const = 0;
  int column_count() const { return schema().attribute_count(); }
  // between one and max_row_count rows, EOS, or an Exception; Next(-1) = "as many as you have" (rowcount_t is unsigned)
  virtual ResultView Next(rowcount_t max_row_count) = 0;
  virtual void Interrupt() = 0;                                     // thread-safe, non-blocking (cursor.h:150-186)
  virtual void AppendDebugDescription(string* target) const = 0;
  virtual bool IsWaitingOnBarrierSupported() const { return false; }
  virtual void ApplyToChildren(CursorTransformer* /*transformer*/) {}   // a device cursor is one fused pipeline: no child cursors to visit
  virtual CursorId GetCursorId() const { return UNKNOWN_ID; }
  string DebugDescription() const { string d; AppendDebugDescription(&d); return d; }
 protected:
  Cursor() {}
 private:
  Cursor(const Cursor&);
  Cursor& operator=(const Cursor&);
};

namespace internal {
// What Operation::CreateCursor() returns: the whole bound operation tree as ONE device plan (ssgpu_plan).
class DeviceCursor : public Cursor {
 public:
  ~DeviceCursor() override { if (res_) ssgpu_result_destroy(res_); if (aux_block_) ssgpu_block_destroy(aux_block_); if (block_) ssgpu_block_destroy(block_); if (plan_) ssgpu_plan_destroy(plan_); }
  const TupleSchema& schema() const override { return schema_; }
  void Interrupt() override { ssgpu_interrupt(plan_); }
  void AppendDebugDescription(string* target) const override { target->append(description_); }
  CursorId GetCursorId() const override { return id_; }

  ResultView Next(rowcount_t max_row_count) override {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    if (id_ == BEST_EFFORT_GROUP_AGGREGATE) {
      // GroupAggregateCursor::Next with best_effort_ (aggregate_groups.cc:211-222): serve the current result; when it has been read and
      // the input is not exhausted, ProcessInput again (ssgpu_plan_run_best_effort).  A view never mixes rows of two results.
      while (!failed_ && (!fetched_ || pos_ >= total_)) {
        if (fetched_ && be_next_row_ >= be_rows_) return ResultView::EOS();
        std::vector<ssgpu_column> cols;
        int rc = StagedColumns(ctx, &cols, &be_rows_);
        if (rc == SSGPU_OK) rc = ssgpu_plan_run_best_effort(plan_, cols.data(), static_cast<int32_t>(cols.size()), be_rows_, be_next_row_, &be_next_row_, &res_);
        if (rc == SSGPU_OK) rc = internal::FetchResult(res_, schema_, &dict_, &total_, &host_data_, &host_null_, &cells_);
        fetched_ = true; ran_ = true; run_rc_ = rc; pos_ = 0;
        if (rc != SSGPU_OK) { failed_ = true; return ResultView::Failure(new Exception(rc, ssgpu_last_error(ctx))); }
      }
    } else if (!fetched_) {
      int rc = RunOnDevice();
      if (rc == SSGPU_OK) rc = internal::FetchResult(res_, schema_, &dict_, &total_, &host_data_, &host_null_, &cells_);
      fetched_ = true;
      if (rc != SSGPU_OK) { failed_ = true; return ResultView::Failure(new Exception(rc, ssgpu_last_error(ctx))); }
    }
    if (failed_) return ResultView::Failure(new Exception(ERROR_UNKNOWN_ERROR, "cursor already failed"));
    if (pos_ >= total_) return ResultView::EOS();
    if (max_row_count == 0) max_row_count = 1;   // (the contract is "between one and max_row_count rows")
    const rowcount_t n = std::min<rowcount_t>(max_row_count, total_ - pos_);
    for (int i = 0; i < schema_.attribute_count(); ++i) {
      const size_t w = SizeOfDataType(schema_.attribute(i).type());
      view_->mutable_column(i)->Reset(static_cast<const char*>(host_data_[i]) + pos_ * w,
                                      host_null_[i] ? reinterpret_cast<const bool*>(host_null_[i]) + pos_ : nullptr);
    }
    view_->set_row_count(n);
    pos_ += n;
    return ResultView::Success(view_.get());
  }

  // Runs the plan and leaves the result in device memory (Next() fetches it; sharded.h packs it into an image instead).
  // Idempotent.  Returns a ReturnCode.
  int RunOnDevice() {
    if (ran_) return run_rc_;
    ran_ = true;
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    const rowcount_t chunk = internal::Context::Get().host_chunk_rows;
    if (!dev_ && !block_ && !aux_ && chunk > 0 && input_->row_count() > chunk && !internal::Dictionary::HasStrings(input_->schema())) {
      // chunked staging straight from the scanned View (the reference drains its child block by block: aggregate_scalar.cc:53-68)
      std::vector<ssgpu_column> cols(static_cast<size_t>(input_->schema().attribute_count()));
      for (size_t i = 0; i < cols.size(); ++i) {
        cols[i].data = input_->column(static_cast<int>(i)).data().raw();
        cols[i].is_null = reinterpret_cast<const uint8_t*>(input_->column(static_cast<int>(i)).is_null());
      }
      int rc = ssgpu_plan_run_host(plan_, cols.data(), static_cast<int32_t>(cols.size()), static_cast<int64_t>(input_->row_count()), static_cast<int64_t>(chunk), &res_);
      if (rc == SSGPU_OK) rc = ssgpu_ctx_synchronize(ctx);   // (the View's memory is the caller's: nothing may read it after Next)
      if (rc != SSGPU_ERROR_NOT_IMPLEMENTED) return run_rc_ = rc;   // (no chunked form for this plan, or a chunk met what only a whole-input run answers: the block path below)
    }
    int rc = Stage(ctx);
    if (rc == SSGPU_OK) {
      if (dev_) rc = ssgpu_plan_run(plan_, dev_->columns.data(), static_cast<int32_t>(dev_->columns.size()), static_cast<int64_t>(dev_->row_count), &res_);
      else rc = ssgpu_plan_run_block(plan_, block_, &res_);
    }
    return run_rc_ = rc;
  }
  ssgpu_plan* plan_handle() const { return plan_; }
  ssgpu_result* result_handle() const { return res_; }

  // A device cursor can run its plan AGAIN (the reference's cursors are single-shot; a sharded job steps the same plans
  // thousands of times and must not rebind them): forget the previous result, keep plan, buffers and staged input.
  // restage = true uploads the host View again (its contents changed); device-resident inputs are never copied.
  void Rewind(bool restage = false) {
    ran_ = false; fetched_ = false; failed_ = false; pos_ = 0; total_ = 0; run_rc_ = SSGPU_OK; be_next_row_ = 0;
    if (restage) { if (block_) { ssgpu_block_destroy(block_); block_ = nullptr; } if (aux_block_) { ssgpu_block_destroy(aux_block_); aux_block_ = nullptr; } }
  }
  // Multi-GPU scalar aggregates (ssgpu.h: partial aggregates): run the shard's rows up to the partial-aggregate state ...
  int RunPartialOnDevice(int64_t global_row_offset) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    int rc = Stage(ctx);
    if (rc != SSGPU_OK) return rc;
    std::vector<ssgpu_column> cols;
    int64_t rows = 0;
    if (dev_) { cols = dev_->columns; rows = static_cast<int64_t>(dev_->row_count); }
    else {
      cols.resize(static_cast<size_t>(input_->schema().attribute_count()));
      for (size_t i = 0; rc == SSGPU_OK && i < cols.size(); ++i) rc = ssgpu_block_column(block_, static_cast<int32_t>(i), &cols[i]);
      rows = ssgpu_block_row_count(block_);
      // the block was staged on the copy stream: the run must wait for it (ssgpu_plan_run_block does this for full runs)
      if (rc == SSGPU_OK) rc = ssgpu_ctx_synchronize(ctx);
    }
    if (rc == SSGPU_OK) rc = ssgpu_plan_run_partial(plan_, cols.data(), static_cast<int32_t>(cols.size()), rows, global_row_offset);
    return rc;
  }
  // Dense-slot GroupAggregate across ranks (ssgpu.h "dense-slot GroupAggregate across ranks"; sharded.h: ShardedGroupAggregate DENSE):
  // the shard's key ranges, the job-wide table layout, a run into the caller's chunked table, the fold of the received chunks.
  int DenseKeyRanges(int32_t* n_keys, uint64_t* lo, uint64_t* hi) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    std::vector<ssgpu_column> cols; int64_t rows = 0;
    int rc = StagedColumns(ctx, &cols, &rows);
    if (rc == SSGPU_OK) rc = ssgpu_plan_key_ranges(plan_, cols.data(), static_cast<int32_t>(cols.size()), rows, n_keys, lo, hi);
    return rc;
  }
  int SetDense(int32_t n_keys, const uint64_t* lo, const uint64_t* hi, int32_t n_chunks, ssgpu_dense_layout* layout) { return ssgpu_plan_set_dense(plan_, n_keys, lo, hi, n_chunks, layout); }
  int RunDense(void* table) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    std::vector<ssgpu_column> cols; int64_t rows = 0;
    int rc = StagedColumns(ctx, &cols, &rows);
    if (rc == SSGPU_OK) rc = ssgpu_plan_run_dense(plan_, cols.data(), static_cast<int32_t>(cols.size()), rows, table);
    return rc;
  }
  int FoldDense(const void* chunks, int32_t n_chunks) {
    const int rc = ssgpu_plan_fold_dense(plan_, chunks, n_chunks, &res_);
    ran_ = true; run_rc_ = rc; fetched_ = false; failed_ = false; pos_ = 0;
    return rc;
  }
  // ... and, after the caller has gathered every rank's state (n_images consecutive copies, device memory), fold and emit.
  int FinalizePartial(const void* gathered_state, int32_t n_images) {
    int rc = ssgpu_plan_fold_finalize(plan_, gathered_state, n_images, &res_);   // fold + state -> slots + emit: one launch
    ran_ = true; run_rc_ = rc; fetched_ = false; failed_ = false; pos_ = 0;
    return rc;
  }

 private:
  friend class BasicOperation;
  DeviceCursor() {}
  // the plan's input as device columns: the caller's device-resident view, or the host View staged into a block (and waited for)
  int StagedColumns(ssgpu_ctx* ctx, std::vector<ssgpu_column>* cols, int64_t* rows) {
    int rc = Stage(ctx);
    if (rc != SSGPU_OK) return rc;
    if (dev_) { *cols = dev_->columns; *rows = static_cast<int64_t>(dev_->row_count); return SSGPU_OK; }
    cols->resize(static_cast<size_t>(input_->schema().attribute_count()));
    for (size_t i = 0; rc == SSGPU_OK && i < cols->size(); ++i) rc = ssgpu_block_column(block_, static_cast<int32_t>(i), &(*cols)[i]);
    *rows = ssgpu_block_row_count(block_);
    if (rc == SSGPU_OK) rc = ssgpu_ctx_synchronize(ctx);   // the block was staged on the copy stream
    return rc;
  }
  int Stage(ssgpu_ctx* ctx) {  // host Views -> device blocks on the copy stream
    if (dev_) return SSGPU_OK;   // device-resident input: nothing to stage
    if (block_) return SSGPU_OK;   // staged by an earlier run of this cursor (Rewind)
    if (aux_) {                // rhs table of a HashJoin: the plan's auxiliary input
      int rc = internal::UploadView(ctx, aux_, &dict_, &aux_block_);
      std::vector<ssgpu_column> cols(aux_->schema().attribute_count());
      for (size_t i = 0; rc == SSGPU_OK && i < cols.size(); ++i) rc = ssgpu_block_column(aux_block_, static_cast<int32_t>(i), &cols[i]);
      if (rc == SSGPU_OK) rc = ssgpu_plan_set_aux_input(plan_, cols.data(), static_cast<int32_t>(cols.size()), static_cast<int64_t>(aux_->row_count()));
      if (rc != SSGPU_OK) return rc;
    }
    return internal::UploadView(ctx, input_, &dict_, &block_);
  }
  ssgpu_plan* plan_ = nullptr;
  ssgpu_block* block_ = nullptr;
  ssgpu_result* res_ = nullptr;
  const View* input_ = nullptr;
  const DeviceView* dev_ = nullptr;   // (this library's own device-resident input: shared with the caller, who may step it -- sharded.h)
  const View* aux_ = nullptr;
  std::unique_ptr<View> input_own_, aux_own_;   // the cursor's copies of the scanned host Views
  ssgpu_block* aux_block_ = nullptr;
  internal::Dictionary dict_;   // STRING cells of the scanned Views and the plan's ConstStrings
  TupleSchema schema_;
  std::unique_ptr<View> view_;
  std::vector<const void*> host_data_;
  std::vector<const uint8_t*> host_null_;
  std::vector<std::vector<StringPiece>> cells_;
  rowcount_t total_ = 0, pos_ = 0;
  int64_t be_next_row_ = 0, be_rows_ = 0;   // BestEffortGroupAggregate: where the next view starts / the input's row count
  bool ran_ = false, failed_ = false, fetched_ = false;
  int run_rc_ = SSGPU_OK;
  CursorId id_ = UNKNOWN_ID;
  string description_;
};
// The device side of a cursor this library made (sharded.h, WriteResultToFile); NULL for a foreign Cursor implementation.
inline DeviceCursor* AsDeviceCursor(Cursor* c) { return dynamic_cast<DeviceCursor*>(c); }
}  // namespace internal

// ---- operations (cursor/base/operation.h:35-83 and the factories of supersonic.h) -------------
// The reference's pure interface (cursor/base/operation.h:35-83).
class Operation {
 public:
  virtual ~Operation() {}
  // The allocator is not owned and must outlive the operation's cursors; NULL resets it to "unset".
  virtual void SetBufferAllocator(BufferAllocator* buffer_allocator, bool cascade_to_children) = 0;
  virtual void SetBufferAllocatorWhereUnset(BufferAllocator* buffer_allocator, bool cascade_to_children) = 0;
  // Binds the whole tree; the Operation must outlive the cursors it returns (operation.h:59).
  virtual FailureOrOwned<Cursor> CreateCursor() const = 0;
  virtual void AppendDebugDescription(string* const target) const = 0;
  string DebugDescription() const { string result; AppendDebugDescription(&result); return result; }
 protected:
  Operation() {}
 private:
  Operation(const Operation&);
  Operation& operator=(const Operation&);
};

namespace internal {
// Every operation the factories below make: a node of the symbolic tree that CreateCursor() flattens into ONE
// ssgpu_plan_desc (the device runs the fused tree; there are no per-operation cursors).
class BasicOperation : public Operation {
 public:
  // Binds the whole tree (Expression::Bind, projector/aggregation binding) and lowers it.
  FailureOrOwned<Cursor> CreateCursor() const override {
    Builder b;
    Emit(&b);
    if (!b.error.empty()) return FailureOrOwned<Cursor>(new Exception(ERROR_NOT_IMPLEMENTED, b.error));
    std::unique_ptr<DeviceCursor> c(new DeviceCursor);
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    // STRING: one order-preserving dictionary over the scanned Views' cells and the plan's ConstStrings
    if (!b.scan && !b.scan_dev) return FailureOrOwned<Cursor>(new Exception(ERROR_INVALID_ARGUMENT_VALUE, "operation tree has no scan"));
    if (!b.string_consts.empty() || (b.scan && internal::Dictionary::HasStrings(b.scan->schema())) || (b.scan_aux && internal::Dictionary::HasStrings(b.scan_aux->schema()))) {
      std::vector<StringPiece> values;
      for (auto& sc : b.string_consts) values.push_back(StringPiece(*sc.second));
      if (b.scan) internal::Dictionary::Collect(*b.scan, &values);
      if (b.scan_aux) internal::Dictionary::Collect(*b.scan_aux, &values);
      int rc = c->dict_.Build(values);
      for (size_t i = 0; rc == SSGPU_OK && i < b.string_consts.size(); ++i) {
        StringPiece v(*b.string_consts[i].second); std::vector<int32_t> code;
        rc = c->dict_.Encode(&v, nullptr, 1, &code);
        b.exprs[static_cast<size_t>(b.string_consts[i].first)].i64 = code[0];
      }
      if (rc != SSGPU_OK) return FailureOrOwned<Cursor>(new Exception(rc, "cannot build the STRING dictionary"));
    }
    std::vector<ssgpu_attr> attrs;
    const TupleSchema& in = b.scan ? b.scan->schema() : b.scan_dev->schema;
    for (int i = 0; i < in.attribute_count(); ++i) attrs.push_back({in.attribute(i).name().c_str(), in.attribute(i).type(), in.attribute(i).nullability()});
    ssgpu_plan_desc d; memset(&d, 0, sizeof(d));
    d.input_schema = attrs.data(); d.n_attrs = static_cast<int32_t>(attrs.size());
    d.ops = b.ops.data(); d.n_ops = static_cast<int32_t>(b.ops.size());
    d.exprs = b.exprs.data(); d.n_exprs = static_cast<int32_t>(b.exprs.size());
    d.expr_args = b.expr_args.data(); d.n_expr_args = static_cast<int32_t>(b.expr_args.size());
    d.projs = b.projs.data(); d.n_projs = static_cast<int32_t>(b.projs.size());
    d.aggs = b.aggs.data(); d.n_aggs = static_cast<int32_t>(b.aggs.size());
    d.sortkeys = b.sortkeys.data(); d.n_sortkeys = static_cast<int32_t>(b.sortkeys.size());
    std::vector<ssgpu_attr> aux_attrs;
    if (b.scan_aux) {
      const TupleSchema& as = b.scan_aux->schema();
      for (int i = 0; i < as.attribute_count(); ++i) aux_attrs.push_back({as.attribute(i).name().c_str(), as.attribute(i).type(), as.attribute(i).nullability()});
      d.aux_schema = aux_attrs.data(); d.n_aux_attrs = static_cast<int32_t>(aux_attrs.size());
    }
    ssgpu_plan* plan = nullptr;
    const int rc = ssgpu_plan_create(ctx, &d, &plan);
    if (rc != SSGPU_OK) return FailureOrOwned<Cursor>(new Exception(rc, ssgpu_last_error(ctx)));
    if (c->dict_.d) ssgpu_plan_set_dict(plan, c->dict_.d);   // CONCAT prints STRING inputs through the cursor's dictionary
    // SetBufferAllocator(MemoryLimit): the plan's device buffers are charged to the allocator's remaining quota
    if (const BufferAllocator* a = EffectiveAllocator()) if (a->has_quota()) ssgpu_plan_set_memory_limit(plan, static_cast<int64_t>(a->Available()));
    // A cursor is self-contained, as the reference's (view_cursor.cc:69-73: the cursor holds its own copy of the View): the
    // Operation tree and the scanned View OBJECTS may die once the cursor exists -- test/guide/primer.cc:225-290 returns a cursor
    // from a function whose View and Operation are locals -- only the arrays the View points at must stay.
    c->plan_ = plan; c->dev_ = b.scan_dev;
    if (b.scan) { c->input_own_.reset(new View(*b.scan)); c->input_ = c->input_own_.get(); }
    if (b.scan_aux) { c->aux_own_.reset(new View(*b.scan_aux)); c->aux_ = c->aux_own_.get(); }
    for (int i = 0; i < ssgpu_plan_attr_count(plan); ++i) {
      ssgpu_attr a; ssgpu_plan_attr(plan, i, &a);
      c->schema_.add_attribute(Attribute(a.name, static_cast<DataType>(a.dtype), static_cast<Nullability>(a.nullable)));
    }
    c->view_.reset(new View(c->schema_));
    c->id_ = cursor_id(); AppendDebugDescription(&c->description_);
    return FailureOrOwned<Cursor>(c.release());
  }
  // The reference's allocator seam (operation.h:48-58).  Device buffers are owned by the plan; an allocator with a
  // quota (MemoryLimit) bounds them, and a run that needs more fails with ERROR_MEMORY_EXCEEDED.  The allocator is not
  // owned and must outlive the cursors.  One plan = one quota: the nearest allocator from the root applies.
  void SetBufferAllocator(BufferAllocator* allocator, bool cascade_to_children) override {
    allocator_ = allocator;
    if (cascade_to_children) for (Operation* child : children()) child->SetBufferAllocator(allocator, true);
  }
  void SetBufferAllocatorWhereUnset(BufferAllocator* allocator, bool cascade_to_children) override {
    if (!allocator_) allocator_ = allocator;
    if (cascade_to_children) for (Operation* child : children()) child->SetBufferAllocatorWhereUnset(allocator, true);
  }
  void AppendDebugDescription(string* const target) const override {
    target->append(name());
    target->append("(");
    bool first = true;
    for (Operation* child : children()) { if (!first) target->append(", "); first = false; child->AppendDebugDescription(target); }
    target->append(")");
  }
  virtual const BufferAllocator* EffectiveAllocator() const { return allocator_; }
  virtual std::vector<Operation*> children() const { return std::vector<Operation*>(); }
  virtual const char* name() const = 0;
  virtual CursorId cursor_id() const { return UNKNOWN_ID; }

  struct Builder {
    std::vector<ssgpu_op> ops; std::vector<ssgpu_expr> exprs; std::vector<int32_t> expr_args;
    std::vector<ssgpu_proj> projs; std::vector<ssgpu_agg> aggs; std::vector<ssgpu_sortkey> sortkeys;
    const View* scan = nullptr;
    const DeviceView* scan_dev = nullptr;   // device-resident input (ScanDeviceView)
    const View* scan_aux = nullptr;   // rhs table of a HashJoin (the plan's auxiliary input)
    bool aux = false;
    std::string error;                // a child that is not one of this library's operations cannot join the fused plan
    std::vector<std::pair<int, const std::string*>> string_consts;   // (expr index, payload): codes are patched in later
    void ProjRange(const std::vector<SingleSourceProjector::Entry>& es, int32_t* first, int32_t* n) {
      *first = static_cast<int32_t>(projs.size()); *n = static_cast<int32_t>(es.size());
      for (auto& e : es) projs.push_back({e.kind, e.position, e.name.c_str(), e.alias.c_str(), e.source, 0});
    }
    int Expr(const Expression* e) {
      std::vector<int32_t> kids;
      for (auto& a : e->args) kids.push_back(Expr(a.get()));
      ssgpu_expr x; memset(&x, 0, sizeof(x));
      x.kind = e->kind; x.op = e->op; x.dtype = e->dtype; x.first_arg = static_cast<int32_t>(expr_args.size()); x.nargs = static_cast<int32_t>(kids.size());
      x.i64 = e->i64; x.f64 = e->f64; x.name = e->name.c_str();
      expr_args.insert(expr_args.end(), kids.begin(), kids.end());
      exprs.push_back(x);
      if (e->kind == SSGPU_EXPR_CONST && e->dtype == STRING) string_consts.push_back({static_cast<int>(exprs.size()) - 1, &e->sval});
      return static_cast<int>(exprs.size()) - 1;
    }
    void Proj(const SingleSourceProjector* p, ssgpu_op* o) {
      o->proj_first = static_cast<int32_t>(projs.size()); o->proj_n = static_cast<int32_t>(p->entries.size());
      for (auto& e : p->entries) projs.push_back({e.kind, e.position, e.name.c_str(), e.alias.c_str(), e.source, 0});
    }
    void Aggs(const AggregationSpecification* s, ssgpu_op* o) {
      o->agg_first = static_cast<int32_t>(aggs.size()); o->agg_n = static_cast<int32_t>(s->elements.size());
      for (auto& e : s->elements) aggs.push_back({e.aggregation, e.distinct ? 1 : 0, e.output_type, 0, e.input.c_str(), e.output.c_str()});
    }
    int Op(ssgpu_op o) { ops.push_back(o); return static_cast<int>(ops.size()) - 1; }
  };
  virtual int Emit(Builder* b) const = 0;
  // a child's node index; a foreign Operation (not made by these factories) has no device form
  static int EmitChild(const Operation* child, Builder* b) {
    const BasicOperation* c = dynamic_cast<const BasicOperation*>(child);
    if (!c) { if (b->error.empty()) b->error = "an operation of the tree is not a device operation of this library: " + (child ? child->DebugDescription() : string("NULL")); return -1; }
    return c->Emit(b);
  }
  static const BufferAllocator* AllocatorOf(const Operation* child) {
    const BasicOperation* c = dynamic_cast<const BasicOperation*>(child);
    return c ? c->EffectiveAllocator() : nullptr;
  }
 protected:
  BufferAllocator* allocator_ = nullptr;
  static ssgpu_op Blank(int kind, int child) { ssgpu_op o; memset(&o, 0, sizeof(o)); o.kind = kind; o.child = child; o.expr = -1; return o; }
};
}  // namespace internal

// ---- BoundExpressionTree (expression/base/expression.h:96-145) over ssgpu_expr_bind / ssgpu_expr_evaluate ------------
// FailureOrReference<const View> (base/exception/result.h): the View stays valid until the next Evaluate.
class EvaluationResult {
 public:
  static EvaluationResult Success(const View* v) { EvaluationResult r; r.view_ = v; return r; }
  static EvaluationResult Failure(Exception* e) { EvaluationResult r; r.exception_.reset(e); return r; }
  bool is_failure() const { return exception_ != nullptr; }
  bool is_success() const { return !is_failure(); }
  const View& get() const { return *view_; }
  const Exception& exception() const { return *exception_; }
 private:
  EvaluationResult() : view_(nullptr) {}
  const View* view_;
  std::shared_ptr<Exception> exception_;
};

// One bool array per column, all of one length (base/infrastructure/bit_pointers.h:541-582): the skip vectors of DoEvaluate.
class BoolView {
 public:
  explicit BoolView(size_t column_count) : columns_(column_count, nullptr), rows_(0) {}
  explicit BoolView(bool_ptr data) : columns_(1, data), rows_(0) {}
  int column_count() const { return static_cast<int>(columns_.size()); }
  rowcount_t row_count() const { return rows_; }
  bool_ptr column(int i) const { return columns_[static_cast<size_t>(i)]; }
  void ResetColumn(int i, bool_ptr data) { columns_[static_cast<size_t>(i)] = data; }
  void set_row_count(rowcount_t rows) { rows_ = rows; }

 private:
  std::vector<bool_ptr> columns_;
  rowcount_t rows_;
};

class BoundExpressionTree {
 public:
  ~BoundExpressionTree() { Release(); }
  const TupleSchema& result_schema() const { return schema_; }
  rowcount_t row_capacity() const { return static_cast<rowcount_t>(ssgpu_expr_row_capacity(plan_)); }
  // One result row per input row, in order.  ERROR_TOO_MANY_ROWS beyond row_capacity() (expression.cc:57-66);
  // evaluation errors (signaling operators) come back as failures.
  EvaluationResult Evaluate(const View& input) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    int rc = SSGPU_OK;
    if (internal::Dictionary::HasStrings(input_schema_)) {   // the constants' codes depend on the View: one dictionary per View
      std::vector<StringPiece> values;
      internal::Dictionary::Collect(input, &values);
      rc = Rebind(ctx, values);
      if (rc != SSGPU_OK) return EvaluationResult::Failure(new Exception(rc, ssgpu_last_error(ctx)));
    }
    if (input.row_count() > row_capacity()) {
      return EvaluationResult::Failure(new Exception(ERROR_TOO_MANY_ROWS, "Trying to evaluate an expression with more rows than its capacity"));
    }
    if (block_) { ssgpu_block_destroy(block_); block_ = nullptr; }
    if (res_) { ssgpu_result_destroy(res_); res_ = nullptr; }
    rc = internal::UploadView(ctx, &input, &dict_, &block_);
    std::vector<ssgpu_column> cols(static_cast<size_t>(input_schema_.attribute_count()));
    for (size_t i = 0; rc == SSGPU_OK && i < cols.size(); ++i) rc = ssgpu_block_column(block_, static_cast<int32_t>(i), &cols[i]);
    if (rc == SSGPU_OK) rc = ssgpu_expr_evaluate(plan_, cols.data(), static_cast<int32_t>(cols.size()), static_cast<int64_t>(input.row_count()), &res_);
    rowcount_t total = 0;
    if (rc == SSGPU_OK) rc = internal::FetchResult(res_, schema_, &dict_, &total, &host_data_, &host_null_, &cells_);
    if (rc != SSGPU_OK) return EvaluationResult::Failure(new Exception(rc, ssgpu_last_error(ctx)));
    for (int i = 0; i < schema_.attribute_count(); ++i)
      view_->mutable_column(i)->Reset(host_data_[static_cast<size_t>(i)], reinterpret_cast<const bool*>(host_null_[static_cast<size_t>(i)]));
    view_->set_row_count(total);
    return EvaluationResult::Success(view_.get());
  }
  // BoundExpression::DoEvaluate(const View& input, const BoolView& skip_vectors) (expression/base/expression.h:46-92): one skip vector per
  // result attribute, in and out -- a row whose byte is set is not evaluated (NULL result, no failure of a signalling operator on it); on
  // return the vector holds the result's NULLs.  A column of the BoolView that is NULL skips nothing (ssgpu_expr_evaluate_skip).
  EvaluationResult DoEvaluate(const View& input, const BoolView& skip_vectors) {
    ssgpu_ctx* ctx = internal::Context::Get().ctx;
    if (internal::Dictionary::HasStrings(input_schema_))
      return EvaluationResult::Failure(new Exception(ERROR_NOT_IMPLEMENTED, "DoEvaluate over STRING attributes: the dictionary is per View (Evaluate)"));
    if (skip_vectors.column_count() != schema_.attribute_count())
      return EvaluationResult::Failure(new Exception(ERROR_ATTRIBUTE_COUNT_MISMATCH, "one skip vector per result attribute"));
    if (input.row_count() > row_capacity())
      return EvaluationResult::Failure(new Exception(ERROR_TOO_MANY_ROWS, "Trying to evaluate an expression with more rows than its capacity"));
    if (block_) { ssgpu_block_destroy(block_); block_ = nullptr; }
    if (res_) { ssgpu_result_destroy(res_); res_ = nullptr; }
    const rowcount_t n = input.row_count();
    int rc = internal::UploadView(ctx, &input, &dict_, &block_);
    // the caller's vectors as a device block of BOOL columns (only the ones it has)
    std::vector<int> have;
    for (int i = 0; i < skip_vectors.column_count(); ++i) if (skip_vectors.column(i)) have.push_back(i);
    ssgpu_block* sblock = nullptr;
    std::vector<std::string> names;
    std::vector<ssgpu_attr> attrs;
    for (size_t k = 0; k < have.size(); ++k) names.push_back("s" + std::to_string(k));
    for (size_t k = 0; k < have.size(); ++k) attrs.push_back({names[k].c_str(), BOOL, NOT_NULLABLE});
    if (rc == SSGPU_OK && !have.empty()) rc = ssgpu_block_create(ctx, attrs.data(), static_cast<int32_t>(attrs.size()), static_cast<int64_t>(std::max<rowcount_t>(n, 1)), &sblock);
    std::vector<uint8_t*> skip(static_cast<size_t>(skip_vectors.column_count()), nullptr);
    for (size_t k = 0; rc == SSGPU_OK && k < have.size(); ++k) {
      if (n > 0) rc = ssgpu_block_upload(sblock, static_cast<int32_t>(k), skip_vectors.column(have[k]), nullptr, 0, static_cast<int64_t>(n));
      ssgpu_column col;
      if (rc == SSGPU_OK) rc = ssgpu_block_column(sblock, static_cast<int32_t>(k), &col);
      if (rc == SSGPU_OK) skip[static_cast<size_t>(have[k])] = static_cast<uint8_t*>(const_cast<void*>(col.data));
    }
    std::vector<ssgpu_column> cols(static_cast<size_t>(input_schema_.attribute_count()));
    for (size_t i = 0; rc == SSGPU_OK && i < cols.size(); ++i) rc = ssgpu_block_column(block_, static_cast<int32_t>(i), &cols[i]);
    ssgpu_result* res = nullptr;     // (owned by the bound tree's skip form inside the library: not destroyed here)
    if (rc == SSGPU_OK) rc = ssgpu_expr_evaluate_skip(plan_, cols.data(), static_cast<int32_t>(cols.size()), static_cast<int64_t>(n), skip.data(), static_cast<int32_t>(skip.size()), &res);
    rowcount_t total = 0;
    TupleSchema nullable_schema;     // every attribute of the skip form's result is NULLABLE
    for (int i = 0; i < schema_.attribute_count(); ++i) nullable_schema.add_attribute(Attribute(schema_.attribute(i).name(), schema_.attribute(i).type(), NULLABLE));
    if (rc == SSGPU_OK) rc = internal::FetchResult(res, nullable_schema, &dict_, &total, &host_data_, &host_null_, &cells_);
    if (sblock) { if (rc == SSGPU_OK) rc = ssgpu_ctx_synchronize(ctx); else (void)ssgpu_ctx_synchronize(ctx); ssgpu_block_destroy(sblock); }
    if (rc != SSGPU_OK) return EvaluationResult::Failure(new Exception(rc, ssgpu_last_error(ctx)));
    if (!skip_view_ || skip_view_->schema().attribute_count() != nullable_schema.attribute_count()) skip_view_.reset(new View(nullable_schema));
    for (int i = 0; i < nullable_schema.attribute_count(); ++i) {
      skip_view_->mutable_column(i)->Reset(host_data_[static_cast<size_t>(i)], reinterpret_cast<const bool*>(host_null_[static_cast<size_t>(i)]));
      if (skip_vectors.column(i) && host_null_[static_cast<size_t>(i)])
        for (rowcount_t r = 0; r < total; ++r) skip_vectors.column(i)[r] = host_null_[static_cast<size_t>(i)][r] != 0;
    }
    skip_view_->set_row_count(total);
    return EvaluationResult::Success(skip_view_.get());
  }

 private:
  friend class Expression;
  BoundExpressionTree() {}
  void Release() {
    if (res_) ssgpu_result_destroy(res_);
    if (block_) ssgpu_block_destroy(block_);
    if (plan_) ssgpu_plan_destroy(plan_);
    res_ = nullptr; block_ = nullptr; plan_ = nullptr;
  }
  // (re)binds the flattened tree against a dictionary of `values` + its own ConstStrings
  int Rebind(ssgpu_ctx* ctx, std::vector<StringPiece> values) {
    Release();
    if (!string_consts_.empty() || !values.empty() || internal::Dictionary::HasStrings(input_schema_)) {
      for (auto& sc : string_consts_) values.push_back(StringPiece(sc.second));
      int rc = dict_.Build(values);
      for (size_t i = 0; rc == SSGPU_OK && i < string_consts_.size(); ++i) {
        StringPiece v(string_consts_[i].second); std::vector<int32_t> code;
        rc = dict_.Encode(&v, nullptr, 1, &code);
        exprs_[static_cast<size_t>(string_consts_[i].first)].i64 = code[0];
      }
      if (rc != SSGPU_OK) return rc;
    }
    std::vector<ssgpu_attr> attrs;
    for (int i = 0; i < input_schema_.attribute_count(); ++i)
      attrs.push_back({input_schema_.attribute(i).name().c_str(), input_schema_.attribute(i).type(), input_schema_.attribute(i).nullability()});
    int rc = ssgpu_expr_bind(ctx, attrs.data(), static_cast<int32_t>(attrs.size()), exprs_.data(), static_cast<int32_t>(exprs_.size()),
                             expr_args_.data(), static_cast<int32_t>(expr_args_.size()), root_, static_cast<int64_t>(max_row_count_), &plan_);
    if (rc != SSGPU_OK) return rc;
    if (memory_limit_ >= 0) ssgpu_plan_set_memory_limit(plan_, memory_limit_);
    schema_ = TupleSchema();
    for (int i = 0; i < ssgpu_plan_attr_count(plan_); ++i) {
      ssgpu_attr a; ssgpu_plan_attr(plan_, i, &a);
      schema_.add_attribute(Attribute(a.name, static_cast<DataType>(a.dtype), static_cast<Nullability>(a.nullable)));
    }
    view_.reset(new View(schema_));
    return SSGPU_OK;
  }
  TupleSchema input_schema_, schema_;
  std::vector<ssgpu_expr> exprs_;                 // the flattened tree; names point into names_
  std::vector<int32_t> expr_args_;
  std::vector<std::unique_ptr<std::string>> names_;
  std::vector<std::pair<int, std::string>> string_consts_;
  int32_t root_ = 0;
  rowcount_t max_row_count_ = 0;
  int64_t memory_limit_ = -1;
  ssgpu_plan* plan_ = nullptr;
  ssgpu_block* block_ = nullptr;
  ssgpu_result* res_ = nullptr;
  internal::Dictionary dict_;
  std::unique_ptr<View> view_;
  std::unique_ptr<View> skip_view_;   // DoEvaluate's result: the same attributes, all NULLABLE
  std::vector<const void*> host_data_;
  std::vector<const uint8_t*> host_null_;
  std::vector<std::vector<StringPiece>> cells_;
};

inline FailureOrOwned<BoundExpressionTree> Expression::Bind(const TupleSchema& input_schema, BufferAllocator* allocator, rowcount_t max_row_count) const {
  std::unique_ptr<BoundExpressionTree> t(new BoundExpressionTree);
  internal::BasicOperation::Builder b;
  t->root_ = b.Expr(this);
  t->exprs_ = b.exprs; t->expr_args_ = b.expr_args;
  for (auto& x : t->exprs_) {            // the bound tree does not depend on this Expression's lifetime
    t->names_.emplace_back(new std::string(x.name ? x.name : ""));
    x.name = t->names_.back()->c_str();
  }
  for (auto& sc : b.string_consts) t->string_consts_.push_back({sc.first, *sc.second});
  t->input_schema_ = input_schema;
  t->max_row_count_ = max_row_count;
  if (allocator && allocator->has_quota()) t->memory_limit_ = static_cast<int64_t>(allocator->Available());
  ssgpu_ctx* ctx = internal::Context::Get().ctx;
  const int rc = t->Rebind(ctx, std::vector<StringPiece>());
  if (rc != SSGPU_OK) return FailureOrOwned<BoundExpressionTree>(new Exception(rc, ssgpu_last_error(ctx)));
  return FailureOrOwned<BoundExpressionTree>(t.release());
}

namespace internal {
class ScanViewOp : public BasicOperation {
 public:
  explicit ScanViewOp(const View& v) : view_(v) {}
  const char* name() const override { return "ScanView"; }
  CursorId cursor_id() const override { return VIEW; }
  int Emit(Builder* b) const override {
    ssgpu_op o = Blank(SSGPU_OP_SCAN, -1);
    if (b->aux) { b->scan_aux = &view_; o.option0 = 1; } else { b->scan = &view_; }
    return b->Op(o);
  }
 private:
  const View& view_;  // must outlive the operation (scan_view.h)
};
class ScanDeviceOp : public BasicOperation {
 public:
  explicit ScanDeviceOp(const DeviceView& v) : view_(v) {}
  const char* name() const override { return "ScanDeviceView"; }
  CursorId cursor_id() const override { return VIEW; }
  int Emit(Builder* b) const override { b->scan_dev = &view_; return b->Op(Blank(SSGPU_OP_SCAN, -1)); }
 private:
  const DeviceView& view_;  // must outlive the operation and its cursors
};
class UnaryOp : public BasicOperation {
 public:
  UnaryOp(int kind, Operation* child, const Expression* e, const SingleSourceProjector* p, const AggregationSpecification* a, const SortOrder* s, int64_t opt = 0)
      : kind_(kind), child_(child), expr_(e), proj_(p), aggs_(a), sort_(s), opt_(opt) {}
  std::vector<Operation*> children() const override { return std::vector<Operation*>(1, child_.get()); }
  const char* name() const override {
    switch (kind_) { case SSGPU_OP_COMPUTE: return "Compute"; case SSGPU_OP_FILTER: return "Filter"; case SSGPU_OP_PROJECT: return "Project";
                     case SSGPU_OP_SCALAR_AGGREGATE: return "ScalarAggregate"; case SSGPU_OP_GROUP_AGGREGATE: return "GroupAggregate";
                     case SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE: return "BestEffortGroupAggregate";
                     case SSGPU_OP_AGGREGATE_CLUSTERS: return "AggregateClusters"; case SSGPU_OP_SORT: return "Sort"; default: return "Operation"; }
  }
  CursorId cursor_id() const override {
    switch (kind_) { case SSGPU_OP_COMPUTE: return COMPUTE; case SSGPU_OP_FILTER: return FILTER; case SSGPU_OP_PROJECT: return PROJECT;
                     case SSGPU_OP_SCALAR_AGGREGATE: return SCALAR_AGGREGATE; case SSGPU_OP_GROUP_AGGREGATE: return GROUP_AGGREGATE;
                     case SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE: return BEST_EFFORT_GROUP_AGGREGATE;
                     case SSGPU_OP_AGGREGATE_CLUSTERS: return AGGREGATE_CLUSTERS; case SSGPU_OP_SORT: return SORT; default: return UNKNOWN_ID; }
  }
  int Emit(Builder* b) const override {
    ssgpu_op o = Blank(kind_, EmitChild(child_.get(), b));
    if (expr_) o.expr = b->Expr(expr_.get());
    if (proj_) b->Proj(proj_.get(), &o);
    if (aggs_) b->Aggs(aggs_.get(), &o);
    if (sort_) { o.sort_first = static_cast<int32_t>(b->sortkeys.size()); o.sort_n = static_cast<int32_t>(sort_->keys.size());
                 for (auto& k : sort_->keys) b->sortkeys.push_back({k.first.c_str(), k.second, 0}); }
    o.option0 = opt_;
    return b->Op(o);
  }
  const BufferAllocator* EffectiveAllocator() const override { return allocator_ ? allocator_ : AllocatorOf(child_.get()); }
 private:
  int kind_;
  std::unique_ptr<Operation> child_;
  std::unique_ptr<const Expression> expr_;
  std::unique_ptr<const SingleSourceProjector> proj_;
  std::unique_ptr<const AggregationSpecification> aggs_;
  std::unique_ptr<const SortOrder> sort_;
  int64_t opt_;
};
// HashJoinOperation (cursor/core/hash_join.h:37-56): INNER / LEFT_OUTER, UNIQUE rhs keys, rhs = ScanView(table)
class HashJoinOp : public BasicOperation {
 public:
  HashJoinOp(JoinType t, const SingleSourceProjector* lk, const SingleSourceProjector* rk, const MultiSourceProjector* rp,
             KeyUniqueness u, Operation* lhs, Operation* rhs) : type_(t), uniq_(u), lk_(lk), rk_(rk), rp_(rp), lhs_(lhs), rhs_(rhs) {}
  std::vector<Operation*> children() const override { std::vector<Operation*> c; c.push_back(lhs_.get()); c.push_back(rhs_.get()); return c; }
  const char* name() const override { return "HashJoin"; }
  CursorId cursor_id() const override { return HASH_JOIN; }
  int Emit(Builder* b) const override {
    const int l = EmitChild(lhs_.get(), b);
    b->aux = true; const int r = EmitChild(rhs_.get(), b); b->aux = false;
    ssgpu_op o = Blank(SSGPU_OP_HASH_JOIN, l);
    o.child2 = r; o.option0 = static_cast<int64_t>(type_) | (static_cast<int64_t>(uniq_) << 8);
    b->ProjRange(lk_->entries, &o.proj_first, &o.proj_n);
    b->ProjRange(rk_->entries, &o.proj2_first, &o.proj2_n);
    b->ProjRange(rp_->entries, &o.proj3_first, &o.proj3_n);
    return b->Op(o);
  }
  const BufferAllocator* EffectiveAllocator() const override { return allocator_ ? allocator_ : AllocatorOf(lhs_.get()); }
 private:
  JoinType type_; KeyUniqueness uniq_;
  std::unique_ptr<const SingleSourceProjector> lk_, rk_;
  std::unique_ptr<const MultiSourceProjector> rp_;
  std::unique_ptr<Operation> lhs_, rhs_;
};
}  // namespace internal

// Takes ownership of all projectors and both children (hash_join.h:48-56).
inline Operation* HashJoin(JoinType join_type, const SingleSourceProjector* lhs_key_selector, const SingleSourceProjector* rhs_key_selector,
                           const MultiSourceProjector* result_projector, KeyUniqueness rhs_key_uniqueness, Operation* lhs_child, Operation* rhs_child) {
  return new internal::HashJoinOp(join_type, lhs_key_selector, rhs_key_selector, result_projector, rhs_key_uniqueness, lhs_child, rhs_child);
}

inline Operation* ScanView(const View& view) { return new internal::ScanViewOp(view); }
inline Operation* ScanDeviceView(const DeviceView& view) { return new internal::ScanDeviceOp(view); }
inline Operation* Compute(const Expression* computation, Operation* child) { return new internal::UnaryOp(SSGPU_OP_COMPUTE, child, computation, nullptr, nullptr, nullptr); }
inline Operation* Project(const SingleSourceProjector* projector, Operation* child) { return new internal::UnaryOp(SSGPU_OP_PROJECT, child, nullptr, projector, nullptr, nullptr); }
inline Operation* Filter(const Expression* predicate, const SingleSourceProjector* projector, Operation* child) { return new internal::UnaryOp(SSGPU_OP_FILTER, child, predicate, projector, nullptr, nullptr); }
inline Operation* ScalarAggregate(AggregationSpecification* spec, Operation* child) { return new internal::UnaryOp(SSGPU_OP_SCALAR_AGGREGATE, child, nullptr, nullptr, spec, nullptr); }
inline Operation* GroupAggregate(const SingleSourceProjector* group_by, const AggregationSpecification* spec, GroupAggregateOptions* options, Operation* child) {
  std::unique_ptr<GroupAggregateOptions> own(options);
  return new internal::UnaryOp(SSGPU_OP_GROUP_AGGREGATE, child, nullptr, group_by, spec, nullptr, options ? options->option0() : 0);
}
// aggregate.h:230-250: groups and aggregates as many input rows as the result block holds, returns them, and starts anew with the
// input it had not consumed -- rows are key-unique within each returned view, not across views; an input of any size is processed
// (ERROR_MEMORY_EXCEEDED is not returned because the input is large).  options->memory_quota bounds the block; without one the
// result is GroupAggregate's.  Takes ownership of its arguments like GroupAggregate.
inline Operation* BestEffortGroupAggregate(const SingleSourceProjector* group_by, const AggregationSpecification* spec, GroupAggregateOptions* options, Operation* child) {
  std::unique_ptr<GroupAggregateOptions> own(options);
  return new internal::UnaryOp(SSGPU_OP_BEST_EFFORT_GROUP_AGGREGATE, child, nullptr, group_by, spec, nullptr, options ? options->memory_quota() : 0);
}
inline Operation* AggregateClusters(const SingleSourceProjector* clustered_by, const AggregationSpecification* spec, Operation* child) { return new internal::UnaryOp(SSGPU_OP_AGGREGATE_CLUSTERS, child, nullptr, clustered_by, spec, nullptr); }
inline Operation* Sort(const SortOrder* order, const SingleSourceProjector* result_projector, size_t memory_limit, Operation* child) {
  return new internal::UnaryOp(SSGPU_OP_SORT, child, nullptr, result_projector ? result_projector : ProjectAllAttributes(), nullptr, order, static_cast<int64_t>(memory_limit));
}

// ---- Arena (base/memory/arena.h): owns the bytes of variable-length values ----------------------------------------------
// StringPiece cells of a View do not own their bytes; user code parks them in an Arena that lives as long as the View
// (test/guide/group_sort.cc:129,273).  The two size arguments of the reference (initial / maximal buffer) have no
// meaning for this implementation and are accepted for source compatibility.
class Arena {
 public:
  Arena(size_t /*initial_buffer_size*/, size_t /*max_buffer_size*/) {}
  Arena(BufferAllocator* /*allocator*/, size_t /*initial_buffer_size*/, size_t /*max_buffer_size*/) {}
  // copies the bytes, returns where they now live (NULL only if memory runs out)
  const char* AddStringPieceContent(const StringPiece& value) { chunks_.emplace_back(value.data(), value.size()); bytes_ += value.size(); return chunks_.back().data(); }
  void Reset() { chunks_.clear(); bytes_ = 0; }
  size_t memory_footprint() const { return bytes_; }
 private:
  std::deque<std::string> chunks_;
  size_t bytes_ = 0;
};

// ---- Block / OwnedColumn (base/infrastructure/block.h:196-286, 412-491; allocation rule block.cc:20-60) ---------------------
// A Block OWNS host columns: per column one buffer of `row_capacity << log2(width)` bytes from the BufferAllocator (pinned
// host memory here, so a Block's view is DMA-able when it is scanned), one byte per row of NULL mask for NULLABLE
// attributes, and an Arena for the bytes of variable-length values.  It has a row capacity and no row count
// (view().row_count() IS the capacity).  User code of the path fills Blocks through ViewCopier (test/guide/group_sort.cc:437-
// 461, join.cc:351-372) or through mutable_column(i)->mutable_typed_data<T>().
class Block;
class OwnedColumn {
 public:
  const Column& content() const { return *column_; }
  void* mutable_data() { return data_ ? data_->data() : nullptr; }
  void* mutable_data_plus_offset(rowcount_t offset) { return static_cast<char*>(mutable_data()) + offset * SizeOfDataType(column_->attribute().type()); }
  template <DataType type> typename TypeTraits<type>::cpp_type* mutable_typed_data() { return static_cast<typename TypeTraits<type>::cpp_type*>(mutable_data()); }
  template <DataType type> typename TypeTraits<type>::cpp_type* mutable_weakly_typed_data() { return static_cast<typename TypeTraits<type>::cpp_type*>(mutable_data()); }
  StringPiece* mutable_variable_length_data() { return static_cast<StringPiece*>(mutable_data()); }
  bool_ptr mutable_is_null() { return is_nullable() && nulls_ ? static_cast<bool*>(nulls_->data()) : nullptr; }   // NULL for NOT_NULLABLE attributes
  bool_ptr mutable_is_null_plus_offset(rowcount_t offset) { bool_ptr z = mutable_is_null(); return z ? z + offset : nullptr; }
  Arena* arena() { return arena_.get(); }                       // variable-length columns only
  // block.cc:20-45: both buffers are (re)allocated, old content is kept up to the smaller capacity
  bool Reallocate(rowcount_t row_capacity, BufferAllocator* allocator) {
    const size_t width = SizeOfDataType(column_->attribute().type());
    if (!Grow(&data_, static_cast<size_t>(row_capacity) * width, allocator)) return false;
    if (is_nullable() && !Grow(&nulls_, static_cast<size_t>(row_capacity), allocator)) return false;
    column_->Reset(data_->data(), is_nullable() ? static_cast<const bool*>(nulls_->data()) : nullptr);
    return true;
  }
 private:
  friend class Block;
  OwnedColumn() : column_(nullptr) {}
  void Init(BufferAllocator* allocator, Column* column) {
    column_ = column;
    const DataType t = column->attribute().type();
    if (t == STRING || t == BINARY) arena_.reset(new Arena(allocator, 0, static_cast<size_t>(-1)));
  }
  bool is_nullable() const { return column_->attribute().is_nullable(); }
  static bool Grow(std::unique_ptr<Buffer>* b, size_t bytes, BufferAllocator* allocator) {
    if (!*b) { b->reset(allocator->Allocate(bytes)); return *b != nullptr; }   // (zero-size requests succeed with non-NULL data, memory.h:112-117)
    return allocator->Reallocate(bytes, b->get());
  }
  Column* column_;                 // the owning Block's view column
  std::unique_ptr<Buffer> data_, nulls_;
  std::unique_ptr<Arena> arena_;
  OwnedColumn(const OwnedColumn&);
  OwnedColumn& operator=(const OwnedColumn&);
};

class Block {
 public:
  Block(const TupleSchema& schema, BufferAllocator* allocator)
      : allocator_(allocator), columns_(new OwnedColumn[static_cast<size_t>(schema.attribute_count() > 0 ? schema.attribute_count() : 1)]), view_(schema) {
    for (int i = 0; i < schema.attribute_count(); ++i) columns_[i].Init(allocator, view_.mutable_column(i));
  }
  // true iff every column got its buffers; on failure the capacity is not raised (block.cc:47-60)
  bool Reallocate(rowcount_t new_row_capacity) {
    if (new_row_capacity < view_.row_count()) view_.set_row_count(new_row_capacity);
    for (int i = 0; i < column_count(); ++i) if (!columns_[i].Reallocate(new_row_capacity, allocator_)) return false;
    view_.set_row_count(new_row_capacity);
    return true;
  }
  void ResetArenas() { for (int i = 0; i < column_count(); ++i) if (columns_[i].arena()) columns_[i].arena()->Reset(); }
  OwnedColumn* mutable_column(int column_index) { return &columns_[column_index]; }
  const View& view() const { return view_; }
  BufferAllocator* allocator() { return allocator_; }
  const TupleSchema& schema() const { return view_.schema(); }
  int column_count() const { return schema().attribute_count(); }
  rowcount_t row_capacity() const { return view_.row_count(); }
  const Column& column(size_t column_index) const { return view_.column(static_cast<int>(column_index)); }
  bool_const_ptr is_null(size_t column_index) const { return column(column_index).is_null(); }
 private:
  BufferAllocator* const allocator_;
  std::unique_ptr<OwnedColumn[]> columns_;
  View view_;
  Block(const Block&);
  Block& operator=(const Block&);
};

// ---- ViewCopier / SelectiveViewCopier (base/infrastructure/view_copier.h:49-158, copy_column.cc:112-240) -------------------
// Copy `row_count` rows of a View into a Block at `output_offset`; with a selector, input row `input_row_ids[i]` goes to
// output row `output_offset + i` and a negative id makes a NULL row (the outer join's missing side, copy_column.cc:199-240).
// deep_copy = variable-length values are copied into the output column's arena (otherwise the cells keep pointing at the
// source's bytes).  Returns the number of rows copied: fewer than asked only when an arena runs out of memory.
enum RowSelectorType { NO_SELECTOR = 0, INPUT_SELECTOR = 1 };
class BaseViewCopier {
 protected:
  BaseViewCopier(const TupleSchema& source_schema, const TupleSchema& result_schema, bool deep_copy) : source_schema_(source_schema), result_schema_(result_schema), deep_copy_(deep_copy) {}
  rowcount_t Copy(const rowcount_t row_count, const View& input_view, const rowid_t* input_row_ids, const rowcount_t output_offset, Block* output_block) const {
    for (int c = 0; c < result_schema_.attribute_count(); ++c) {
      const Column& in = input_view.column(c);
      OwnedColumn* out = output_block->mutable_column(c);
      const DataType type = result_schema_.attribute(c).type();
      const size_t width = SizeOfDataType(type);
      const bool variable = type == STRING || type == BINARY;
      bool_const_ptr in_null = in.is_null();
      bool_ptr out_null = out->mutable_is_null_plus_offset(output_offset);
      const char* src = static_cast<const char*>(in.data().raw());
      char* dst = static_cast<char*>(out->mutable_data_plus_offset(output_offset));
      if (!input_row_ids && !variable) {   // a run of whole rows: one copy per buffer
        if (row_count) memcpy(dst, src, static_cast<size_t>(row_count) * width);
        if (out_null) { if (in_null) memcpy(out_null, in_null, static_cast<size_t>(row_count)); else if (row_count) memset(out_null, 0, static_cast<size_t>(row_count)); }
        continue;
      }
      for (rowcount_t i = 0; i < row_count; ++i) {
        const rowid_t r = input_row_ids ? input_row_ids[i] : static_cast<rowid_t>(i);
        const bool is_null = r < 0 || (in_null && in_null[r]);
        if (out_null) out_null[i] = is_null;
        if (is_null) { if (variable) reinterpret_cast<StringPiece*>(dst)[i] = StringPiece(); continue; }
        if (variable) {
          const StringPiece& cell = reinterpret_cast<const StringPiece*>(src)[r];
          if (deep_copy_) {
            const char* kept = out->arena()->AddStringPieceContent(cell);
            if (kept == nullptr) return i;
            reinterpret_cast<StringPiece*>(dst)[i] = StringPiece(kept, cell.size());
          } else {
            reinterpret_cast<StringPiece*>(dst)[i] = cell;
          }
        } else {
          memcpy(dst + i * width, src + static_cast<size_t>(r) * width, width);
        }
      }
    }
    return row_count;
  }
 private:
  TupleSchema source_schema_, result_schema_;
  bool deep_copy_;
};
class ViewCopier : public BaseViewCopier {
 public:
  ViewCopier(const TupleSchema& schema, bool deep_copy) : BaseViewCopier(schema, schema, deep_copy) {}
  rowcount_t Copy(const rowcount_t row_count, const View& input_view, const rowcount_t output_offset, Block* output_block) const {
    return BaseViewCopier::Copy(row_count, input_view, nullptr, output_offset, output_block);
  }
};
class SelectiveViewCopier : public BaseViewCopier {
 public:
  SelectiveViewCopier(const TupleSchema& schema, bool deep_copy) : BaseViewCopier(schema, schema, deep_copy) {}
  SelectiveViewCopier(const TupleSchema& source_schema, const TupleSchema& result_schema, bool deep_copy) : BaseViewCopier(source_schema, result_schema, deep_copy) {}
  rowcount_t Copy(const rowcount_t row_count, const View& input_view, const rowid_t* input_row_ids, const rowcount_t output_offset, Block* output_block) const {
    return BaseViewCopier::Copy(row_count, input_view, input_row_ids, output_offset, output_block);
  }
};

// ---- Table / TableRowWriter (cursor/infrastructure/table.h:49-290) --------------------------------------------------------
// An Operation that owns a growable block of rows on the HOST and scans it: the container user code fills row by row
// (TableRowWriter) or view by view (AppendView) and then hands to an operation tree like any other child.  Column
// buffers come from the BufferAllocator given (pinned host memory: the scan's upload is a DMA), variable-length values
// are deep-copied into the Table's own arena (the reference's rule, table.cc / view_copier.h).
class Table : public internal::BasicOperation {
 public:
  Table(const TupleSchema& schema, BufferAllocator* buffer_allocator)
      : alloc_(buffer_allocator ? buffer_allocator : HeapBufferAllocator::Get()), view_(schema), arena_(0, 0),
        data_(static_cast<size_t>(schema.attribute_count())), nulls_(static_cast<size_t>(schema.attribute_count())), capacity_(0) {}
  ~Table() override {}
  const char* name() const override { return "Table"; }
  CursorId cursor_id() const override { return VIEW; }
  int Emit(Builder* b) const override {   // = ScanView(view())
    ssgpu_op o = Blank(SSGPU_OP_SCAN, -1);
    if (b->aux) { b->scan_aux = &view_; o.option0 = 1; } else { b->scan = &view_; }
    return b->Op(o);
  }
  const View& view() const { return view_; }
  const TupleSchema& schema() const { return view_.schema(); }
  rowcount_t row_count() const { return view_.row_count(); }
  rowcount_t row_capacity() const { return capacity_; }
  void Clear() { view_.set_row_count(0); arena_.Reset(); }
  // Capacity for at least `needed_capacity` rows; false (and the old capacity) if the allocator refuses.
  bool ReserveRowCapacity(rowcount_t needed_capacity) {
    if (needed_capacity <= capacity_) return true;
    rowcount_t cap = capacity_ ? capacity_ : 16;
    while (cap < needed_capacity) cap *= 2;
    return SetRowCapacity(cap);
  }
  bool SetRowCapacity(rowcount_t row_capacity) {
    if (row_capacity < row_count()) return false;
    // allocate all new buffers first: on failure nothing has changed
    std::vector<std::unique_ptr<Buffer>> nd(data_.size()), nn(nulls_.size());
    for (int i = 0; i < schema().attribute_count(); ++i) {
      const size_t w = SizeOfDataType(schema().attribute(i).type());
      nd[static_cast<size_t>(i)].reset(alloc_->Allocate(static_cast<size_t>(row_capacity) * w));
      if (!nd[static_cast<size_t>(i)]) return false;
      if (schema().attribute(i).is_nullable()) {
        nn[static_cast<size_t>(i)].reset(alloc_->Allocate(static_cast<size_t>(row_capacity)));
        if (!nn[static_cast<size_t>(i)]) return false;
      }
    }
    for (int i = 0; i < schema().attribute_count(); ++i) {
      const size_t k = static_cast<size_t>(i), w = SizeOfDataType(schema().attribute(i).type());
      if (row_count()) memcpy(nd[k]->data(), data_[k]->data(), static_cast<size_t>(row_count()) * w);
      if (nn[k]) { memset(nn[k]->data(), 0, static_cast<size_t>(row_capacity)); if (row_count()) memcpy(nn[k]->data(), nulls_[k]->data(), static_cast<size_t>(row_count())); }
      data_[k] = std::move(nd[k]); nulls_[k] = std::move(nn[k]);
      view_.mutable_column(i)->Reset(data_[k]->data(), nulls_[k] ? static_cast<const bool*>(nulls_[k]->data()) : nullptr);
    }
    capacity_ = row_capacity;
    return true;
  }
  // A new row (its cells unset: the caller Sets / SetNulls every one of them); its index, or -1 when out of memory.
  rowid_t AddRow() {
    if (!ReserveRowCapacity(row_count() + 1)) return -1;
    view_.set_row_count(row_count() + 1);
    return static_cast<rowid_t>(row_count() - 1);
  }
  template <DataType type> bool Set(int col_index, rowid_t row_index, const typename TypeTraits<type>::cpp_type& value) {
    typedef typename TypeTraits<type>::cpp_type T;
    if (schema().attribute(col_index).type() != type) return false;
    const size_t k = static_cast<size_t>(col_index);
    static_cast<T*>(data_[k]->data())[row_index] = Keep(value);
    if (nulls_[k]) static_cast<bool*>(nulls_[k]->data())[row_index] = false;
    return true;
  }
  void SetNull(int col_index, rowid_t row_index) {
    const size_t k = static_cast<size_t>(col_index);
    if (nulls_[k]) static_cast<bool*>(nulls_[k]->data())[row_index] = true;
  }
  // Appends (deep-copies) the rows of a View with this Table's schema; returns the number of rows appended.
  rowcount_t AppendView(const View& view) {
    const rowcount_t n = view.row_count(), at = row_count();
    if (n == 0 || !ReserveRowCapacity(at + n)) return 0;
    for (int i = 0; i < schema().attribute_count(); ++i) {
      const size_t k = static_cast<size_t>(i), w = SizeOfDataType(schema().attribute(i).type());
      const bool* z = view.column(i).is_null();
      if (schema().attribute(i).type() == STRING || schema().attribute(i).type() == BINARY) {
        const StringPiece* src = view.column(i).variable_length_data();
        StringPiece* dst = static_cast<StringPiece*>(data_[k]->data()) + at;
        for (rowcount_t r = 0; r < n; ++r) dst[r] = (z && z[r]) ? StringPiece() : Keep(src[r]);
      } else {
        memcpy(static_cast<char*>(data_[k]->data()) + at * w, view.column(i).data().raw(), static_cast<size_t>(n) * w);
      }
      if (nulls_[k]) { bool* dz = static_cast<bool*>(nulls_[k]->data()) + at; for (rowcount_t r = 0; r < n; ++r) dz[r] = z ? z[r] : false; }
    }
    view_.set_row_count(at + n);
    return n;
  }
  bool CopyFrom(const Table& other) { Clear(); return AppendView(other.view()) == other.view().row_count(); }
 private:
  template <typename T> const T& Keep(const T& v) { return v; }
  StringPiece Keep(const StringPiece& v) { return StringPiece(arena_.AddStringPieceContent(v), v.size()); }   // deep copy
  BufferAllocator* alloc_;
  View view_;
  Arena arena_;
  std::vector<std::unique_ptr<Buffer>> data_, nulls_;
  rowcount_t capacity_;
};

// table.h:212-290: `writer.AddRow().Int32(1).String("x").Null()` ... `writer.CheckSuccess()`
class TableRowWriter {
 public:
  explicit TableRowWriter(Table* table) : table_(table), col_index_(table->schema().attribute_count()), row_index_(-1), failed_(false) {}
  TableRowWriter& AddRow() {
    if (success()) {
      if (col_index_ != table_->schema().attribute_count()) { failed_ = true; return *this; }   // the previous row is not complete
      row_index_ = table_->AddRow();
      failed_ = row_index_ < 0;
      col_index_ = 0;
    }
    return *this;
  }
  TableRowWriter& Int32(int32 value) { return Set<INT32>(value); }
  TableRowWriter& Int64(int64 value) { return Set<INT64>(value); }
  TableRowWriter& Uint32(uint32 value) { return Set<UINT32>(value); }
  TableRowWriter& Uint64(uint64 value) { return Set<UINT64>(value); }
  TableRowWriter& Float(float value) { return Set<FLOAT>(value); }
  TableRowWriter& Double(double value) { return Set<DOUBLE>(value); }
  TableRowWriter& Bool(bool value) { return Set<BOOL>(value); }
  TableRowWriter& Date(int32 value) { return Set<DATE>(value); }
  TableRowWriter& Datetime(int64 value) { return Set<DATETIME>(value); }
  TableRowWriter& String(const StringPiece& value) { return Set<STRING>(value); }
  TableRowWriter& Null() {
    if (success()) { if (col_index_ >= table_->schema().attribute_count()) failed_ = true; else table_->SetNull(col_index_++, row_index_); }
    return *this;
  }
  TableRowWriter& AllFurtherNull() {
    if (success()) while (col_index_ < table_->schema().attribute_count()) table_->SetNull(col_index_++, row_index_);
    return *this;
  }
  template <DataType type> TableRowWriter& Set(const typename TypeTraits<type>::cpp_type& value) {
    if (success()) failed_ = col_index_ >= table_->schema().attribute_count() || !table_->Set<type>(col_index_++, row_index_, value);
    return *this;
  }
  bool success() const { return !failed_; }
  void CheckSuccess() const {
    if (!success()) { fprintf(stderr, "TableRowWriter failed at row %lld, column %d\n", static_cast<long long>(row_index_), col_index_); abort(); }
  }
  const TupleSchema& schema() const { return table_->schema(); }
 private:
  Table* table_;
  int col_index_;
  rowid_t row_index_;
  bool failed_;
};

// ---- the View file format (cursor/infrastructure/file_io.h:58-72, file_io.cc:15-30) ---------------------------------
// The reference hands its own File* to these; the mirror takes a path (the ABI reads with positional readers into pinned
// slabs, which a File* abstraction cannot give it).  Chunks of <= 8192 rows: a uint64 row count, then per column the
// is_null bytes (NULLABLE attributes) and the data -- raw for fixed-width types, a uint64 length per row (0 for NULL
// and empty) followed by one run of bytes for STRING.

// FileOutput(file, ownership) -> Sink (cursor.h Sink: Write(view) -> rows written, Finalize()).
class Sink {
 public:
  virtual ~Sink() {}
  virtual FailureOr<rowcount_t> Write(const View& data) = 0;
  virtual FailureOrVoid Finalize() = 0;
};

namespace internal {
class FileSink : public Sink {
 public:
  explicit FileSink(const std::string& path) : f_(fopen(path.c_str(), "wb")) {}
  ~FileSink() override { if (f_) fclose(f_); }
  FailureOr<rowcount_t> Write(const View& v) override {
    static const rowcount_t kMaxChunkRowCount = 8192;   // file_io.cc:70
    if (!f_) return FailureOr<rowcount_t>(new Exception(ERROR_GENERAL_IO_ERROR, "Writing view to the output file failed."));
    bool ok = true;
    auto put = [&](const void* p, size_t n) { ok = ok && (n == 0 || fwrite(p, 1, n, f_) == n); };
    for (rowcount_t off = 0; off < v.row_count(); off += kMaxChunkRowCount) {
      const uint64_t rc = std::min<rowcount_t>(kMaxChunkRowCount, v.row_count() - off);
      put(&rc, 8);
      for (int i = 0; i < v.column_count(); ++i) {
        const Attribute& a = v.schema().attribute(i);
        const bool* nulls = v.column(i).is_null();
        if (a.is_nullable()) {
          if (nulls) put(nulls + off, rc);
          else { std::vector<char> zeros(rc, 0); put(zeros.data(), rc); }
        }
        if (a.type() == STRING || a.type() == BINARY) {
          const StringPiece* cells = v.column(i).typed_data<StringPiece>() + off;
          std::vector<uint64_t> lens(rc);
          for (uint64_t r = 0; r < rc; ++r) lens[r] = (a.is_nullable() && nulls && nulls[off + r]) ? 0 : cells[r].size();
          put(lens.data(), rc * 8);
          for (uint64_t r = 0; r < rc; ++r) put(cells[r].data(), lens[r]);
        } else {
          put(static_cast<const char*>(v.column(i).data().raw()) + off * SizeOfDataType(a.type()), rc * SizeOfDataType(a.type()));
        }
      }
    }
    if (!ok) return FailureOr<rowcount_t>(new Exception(ERROR_GENERAL_IO_ERROR, "Writing view to the output file failed."));
    return FailureOr<rowcount_t>(v.row_count());
  }
  FailureOrVoid Finalize() override {
    const bool ok = !f_ || fclose(f_) == 0;
    f_ = nullptr;
    return ok ? FailureOrVoid() : FailureOrVoid(new Exception(ERROR_GENERAL_IO_ERROR, "Error closing the file."));
  }
 private:
  FILE* f_;
};
}  // namespace internal

inline Sink* FileOutput(const std::string& path) { return new internal::FileSink(path); }

// A finished cursor's result straight from device memory into a file (fixed-width columns): FileOutput(...)->Write of
// the whole result without the host View in between.
inline FailureOrVoid WriteResultToFile(Cursor* cursor, const std::string& path) {
  internal::DeviceCursor* dc = internal::AsDeviceCursor(cursor);
  if (!dc) return FailureOrVoid(new Exception(ERROR_NOT_IMPLEMENTED, "WriteResultToFile needs a cursor made by this library's operations"));
  int rc = dc->RunOnDevice();
  if (rc == SSGPU_OK) rc = ssgpu_result_write_file(dc->result_handle(), path.c_str());
  return rc == SSGPU_OK ? FailureOrVoid() : FailureOrVoid(new Exception(rc, ssgpu_last_error(internal::Context::Get().ctx)));
}

// FileInput(schema, file, delete_when_done, allocator) drained into device memory: the columns of the whole file as a
// DeviceView (scan it with ScanDeviceView).  Owns the device block.  Fixed-width columns only: STRING columns have to meet
// a plan's dictionary on the host first (read them with the host tools, then ScanView).
class DeviceTable {
 public:
  ~DeviceTable() { if (block_) ssgpu_block_destroy(block_); }
  const DeviceView& view() const { return view_; }
  rowcount_t row_count() const { return view_.row_count; }
 private:
  friend FailureOrOwned<DeviceTable> FileInput(const TupleSchema&, const std::string&, bool);
  DeviceTable() {}
  ssgpu_block* block_ = nullptr;
  DeviceView view_;
};

inline FailureOrOwned<DeviceTable> FileInput(const TupleSchema& schema, const std::string& path, bool delete_when_done = false) {
  ssgpu_ctx* ctx = internal::Context::Get().ctx;
  std::vector<ssgpu_attr> attrs(schema.attribute_count());
  for (int i = 0; i < schema.attribute_count(); ++i) {
    attrs[i].name = schema.attribute(i).name().c_str();
    attrs[i].dtype = schema.attribute(i).type();
    attrs[i].nullable = schema.attribute(i).is_nullable() ? 1 : 0;
  }
  std::unique_ptr<DeviceTable> t(new DeviceTable);
  int rc = ssgpu_block_create_from_file(ctx, attrs.data(), static_cast<int32_t>(attrs.size()), path.c_str(), &t->block_);
  if (rc != SSGPU_OK) return FailureOrOwned<DeviceTable>(new Exception(rc, ssgpu_last_error(ctx)));
  t->view_.schema = schema;
  t->view_.columns.resize(attrs.size());
  for (size_t i = 0; i < attrs.size(); ++i) ssgpu_block_column(t->block_, static_cast<int32_t>(i), &t->view_.columns[i]);
  t->view_.row_count = ssgpu_block_row_count(t->block_);
  if (delete_when_done) remove(path.c_str());
  return FailureOrOwned<DeviceTable>(t.release());
}

}  // namespace supersonic
#endif  // SUPERSONIC_AMD_SUPERSONIC_H_
