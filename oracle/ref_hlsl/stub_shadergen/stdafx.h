// stdafx.h — stand-in for the reference's Windows precompiled header (Source/stdafx.h pulls in ATL, DirectShow and D3D9).
// TEST INFRASTRUCTURE ONLY; ours.  Only what Source/Shaders.cpp touches: a handful of Win32 typedefs, a no-op DLog, and a
// std::format for the "{}" patterns that file uses (libstdc++ 11 has no <format>).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <string>
#include <charconv>
#include <algorithm>
#include <type_traits>

typedef int32_t  HRESULT;
typedef int32_t  LONG;
typedef uint32_t UINT;
typedef uint32_t DWORD;
typedef uint32_t ULONG;
typedef unsigned char BYTE;
typedef int      BOOL;
typedef const char* LPCSTR;
typedef const wchar_t* LPCWSTR;
typedef void*    LPVOID;
typedef void*    HMODULE;
struct RECT { LONG left, top, right, bottom; };
struct GUID { uint32_t a; uint16_t b, c; uint8_t d[8]; };
struct IUnknown { };
#define interface struct
#define __declspec(x)
#define STDMETHOD(m) virtual HRESULT m
#define PURE = 0
#define DEFINE_GUID(name, ...) static const GUID name = {}
#define S_OK   ((HRESULT)0)
#define E_FAIL ((HRESULT)0x80004005)
#define FAILED(hr) (((HRESULT)(hr)) < 0)
#define ASSERT(x) ((void)0)
#define SAFE_RELEASE(p) do { if (p) { (p)->Release(); (p) = nullptr; } } while (0)
template <class... A> inline void DLog(A&&...) {}

HMODULE LoadLibraryW(const wchar_t* name);                 // ref_shadergen_shim.cpp: hands back the text-capturing "compiler"
void*   GetProcAddress(HMODULE h, const char* name);
HRESULT GetDataFromResource(LPVOID& data, DWORD& size, UINT resid);   // ... and the .hlsl files as resources

namespace std {
namespace fmtshim {
inline void put(string& o, const char* s) { o += s; }
inline void put(string& o, const string& s) { o += s; }
inline void put(string& o, float v) { char b[64]; auto r = to_chars(b, b + sizeof b, v); o.append(b, r.ptr); }   // shortest round trip, as std::format
inline void put(string& o, double v) { char b[64]; auto r = to_chars(b, b + sizeof b, v); o.append(b, r.ptr); }
template <class T> inline typename enable_if<is_integral<T>::value>::type put(string& o, T v) { o += to_string(v); }
inline void walk(string& o, const char*& f)
{
    for (; *f; f++) {
        if (f[0] == '{' && f[1] == '{') { o += '{'; f++; }
        else if (f[0] == '}' && f[1] == '}') { o += '}'; f++; }
        else o += *f;
    }
}
template <class T, class... R> inline void walk(string& o, const char*& f, const T& t, const R&... r)
{
    for (; *f; f++) {
        if (f[0] == '{' && f[1] == '{') { o += '{'; f++; }
        else if (f[0] == '}' && f[1] == '}') { o += '}'; f++; }
        else if (f[0] == '{' && f[1] == '}') { put(o, t); f += 2; walk(o, f, r...); return; }
        else o += *f;
    }
}
}  // namespace fmtshim
template <class... A> inline string format(const char* f, const A&... a) { string o; fmtshim::walk(o, f, a...); return o; }
}  // namespace std
