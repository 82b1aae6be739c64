// tests/adapter_env/win32_standins.h — TEST INFRASTRUCTURE ONLY (ours): the Windows / DirectShow / ATL names the reference's CVideoProcessor
// interface (Source/VideoProcessor.h:171-236) mentions, as opaque stand-ins, so that examples/hip_video_processor_adapter.cpp can be
// compiled here against the REAL method declarations (cut out of the reference header at test time) and the REAL Settings_t
// (Source/IVideoRenderer.h, included as it is) — and, since round 6, linked with a driver and run (build_adapter.py, adapter_driver.cpp).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>

typedef int32_t HRESULT;
typedef int32_t LONG;
typedef uint32_t UINT;
typedef uint32_t DWORD;
typedef uint32_t ULONG;
typedef unsigned char BYTE;
typedef int BOOL;
typedef int64_t REFERENCE_TIME;
typedef void *HWND;
struct SIZE { LONG cx, cy; };
struct RECT { LONG left, top, right, bottom; };
struct CRect : RECT {
    CRect() : RECT{0, 0, 0, 0} {}
    CRect(LONG l, LONG t, LONG r, LONG b) : RECT{l, t, r, b} {}
    LONG Width() const { return right - left; }
    LONG Height() const { return bottom - top; }
};
#define S_OK ((HRESULT)0)
#define S_FALSE ((HRESULT)1)
#define E_NOTIMPL ((HRESULT)0x80004001)
#define E_FAIL ((HRESULT)0x80004005)
#define E_POINTER ((HRESULT)0x80004003)
#define E_OUTOFMEMORY ((HRESULT)0x8007000E)
#define LMEM_FIXED 0
inline void *LocalAlloc(unsigned, size_t n) { return std::malloc(n); }
inline void *LocalFree(void *p) { std::free(p); return nullptr; }
#define SUCCEEDED(hr) (((HRESULT)(hr)) >= 0)
#define FAILED(hr) (((HRESULT)(hr)) < 0)
#define AMCONTROL_USED 0x00000001u
#define AMCONTROL_COLORINFO_PRESENT 0x00000080u

struct GUID { uint32_t a; uint16_t b, c; uint8_t d[8]; };
#define DEFINE_GUID(name, ...) static const GUID name = {}
#define interface struct
struct IUnknown { };
#define STDMETHOD(m) virtual HRESULT m
#define STDMETHOD_(t, m) virtual t m
#define PURE = 0
#define __declspec(x)
struct IDirect3DDeviceManager9;
struct ISubPicAllocator;
struct DisplayConfig_t;
struct BITMAPINFOHEADER { DWORD biSize; LONG biWidth, biHeight; uint16_t biPlanes, biBitCount; DWORD biCompression, biSizeImage; LONG biXPelsPerMeter, biYPelsPerMeter; DWORD biClrUsed, biClrImportant; };
struct VIDEOINFOHEADER2 { RECT rcSource, rcTarget; DWORD dwBitRate, dwBitErrorRate; REFERENCE_TIME AvgTimePerFrame; DWORD dwInterlaceFlags, dwCopyProtectFlags, dwPictAspectRatioX, dwPictAspectRatioY, dwControlFlags, dwReserved2; BITMAPINFOHEADER bmiHeader; };
struct CMediaType { BYTE *pbFormat; };
struct IMediaSample {
    virtual HRESULT GetPointer(BYTE **pp) = 0;
    virtual long GetActualDataLength() = 0;
};
// side data of a sample (Include/IMediaSideData.h declares the real interface; only the one call the adapter makes)

// Source/Helper.h: ColorFormat_t (Helper.h:86-127, same order) and the fields of FmtConvParams_t the adapter reads
enum ColorFormat_t {
    CF_NONE = 0, CF_NV12, CF_P010, CF_P016, CF_YUY2, CF_UYVY, CF_P210, CF_P216, CF_Y210, CF_Y216, CF_V210, CF_AYUV, CF_Y410, CF_Y416,
    CF_YV12, CF_YV16, CF_YV24, CF_YUV420P8, CF_YUV422P8, CF_YUV444P8, CF_YUV420P10, CF_YUV420P16, CF_YUV422P10, CF_YUV422P16,
    CF_YUV444P10, CF_YUV444P16, CF_GBRP8, CF_GBRP10, CF_GBRP16, CF_RGB24, CF_XRGB32, CF_ARGB32, CF_r210, CF_RGB48, CF_BGR48,
    CF_BGRA64, CF_B64A, CF_Y8, CF_Y10, CF_Y16,
};
struct FmtConvParams_t { ColorFormat_t cformat; const wchar_t *str; int Packsize, PitchCoeff, Subsampling, CDepth; };
const FmtConvParams_t &GetFmtConvParams(const CMediaType *pmt);
const FmtConvParams_t &GetFmtConvParams(ColorFormat_t fmt);
