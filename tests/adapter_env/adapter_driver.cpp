// tests/adapter_env/adapter_driver.cpp — TEST INFRASTRUCTURE: one frame through the adapter CLASS
// (examples/hip_video_processor_adapter.cpp, compiled against the reference's own CVideoProcessor method list) the way the filter drives
// a video processor (VideoRenderer.cpp:285-287, 424, 447-479, 504, 990): construct -> Init -> VerifyMediaType / InitMediaType ->
// SetWindowRect / SetVideoRect -> ProcessSample(IMediaSample) [= CopySample + Render] -> GetCurentImage -> GetDisplayedImage -> GetVPInfo,
// every call through the base-class pointer.  Built by build_adapter.py into oracle/_ref/adapter_driver (needs /root/reference), run on
// the GPU box by tests/test_parity_gpu.py::test_one_frame_through_the_adapter_class, which compares what comes back with the oracle.
//
//   adapter_driver <cformat> <w> <h> <extfmt> <win_w> <win_h> <vl> <vt> <vr> <vb> <iUpscaling> <ten_bit 0|1> <sample.bin> <displayed.bin> <snapshot.bin>
#include <cstdio>
#include <vector>

#include "VideoProcessor.h"

CVideoProcessor *CreateHipVideoProcessor(CMpcVideoRenderer *pFilter, const Settings_t &config, HRESULT &hr);

// ---- what the filter side provides (stand-ins for Helper.cpp / VideoProcessor.cpp, which need the Windows SDK) ----
static FmtConvParams_t g_fmt{};
const FmtConvParams_t &GetFmtConvParams(const CMediaType *) { return g_fmt; }               // Helper.cpp: subtype GUID -> s_FmtConvMapping row
const FmtConvParams_t &GetFmtConvParams(ColorFormat_t) { return g_fmt; }
bool CVideoProcessor::CheckDoviMetadata(const MediaSideDataDOVIMetadata *, const uint8_t) { return true; }

struct FakeSample : IMediaSample {          // a media sample over a host buffer (the decoder's frame)
    std::vector<BYTE> data;
    HRESULT GetPointer(BYTE **pp) override { *pp = data.data(); return S_OK; }
    long GetActualDataLength() override { return (long)data.size(); }
};

static bool write_file(const char *path, const void *p, size_t n)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(p, 1, n, f) == n;
    std::fclose(f);
    return ok;
}
#define CHECK(call) do { const HRESULT hr_ = (call); if (FAILED(hr_)) { std::fprintf(stderr, "%s -> %08x\n", #call, (unsigned)hr_); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 16) { std::fprintf(stderr, "usage: see the header of adapter_driver.cpp\n"); return 64; }
    const int cformat = std::atoi(argv[1]), w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const unsigned extfmt = (unsigned)std::strtoul(argv[4], nullptr, 0);
    const int win_w = std::atoi(argv[5]), win_h = std::atoi(argv[6]);
    const CRect video(std::atoi(argv[7]), std::atoi(argv[8]), std::atoi(argv[9]), std::atoi(argv[10]));
    Settings_t sets;                                 // the reference's own defaults (IVideoRenderer.h)
    sets.iUpscaling = std::atoi(argv[11]);
    const bool ten = std::atoi(argv[12]) != 0;
    (void)ten;

    FakeSample sample;
    {
        FILE *f = std::fopen(argv[13], "rb");
        if (!f) { std::perror(argv[13]); return 66; }
        std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        sample.data.resize((size_t)n);
        if (std::fread(sample.data.data(), 1, (size_t)n, f) != (size_t)n) return 66;
        std::fclose(f);
    }

    HRESULT hr = E_FAIL;
    CVideoProcessor *vp = CreateHipVideoProcessor(nullptr, sets, hr);      // VideoRenderer.cpp:285-287
    CHECK(hr);
    CHECK(vp->Init((HWND)0x1234, false));
    if (!vp->IsInit()) return 3;

    g_fmt.cformat = (ColorFormat_t)cformat;
    VIDEOINFOHEADER2 vih2{};
    vih2.rcSource = RECT{0, 0, w, h};
    vih2.bmiHeader.biWidth = w; vih2.bmiHeader.biHeight = h;
    vih2.dwControlFlags = extfmt;                    // DXVA2_ExtendedFormat bits with AMCONTROL_USED | AMCONTROL_COLORINFO_PRESENT (the decoder sets them)
    CMediaType mt{(BYTE *)&vih2};
    if (!vp->VerifyMediaType(&mt)) { std::fprintf(stderr, "VerifyMediaType refused the format\n"); return 4; }
    if (!vp->InitMediaType(&mt)) { std::fprintf(stderr, "InitMediaType failed\n"); return 4; }
    CHECK(vp->SetWindowRect(CRect(0, 0, win_w, win_h)));
    vp->SetVideoRect(video);

    CHECK(vp->ProcessSample(&sample));               // CopySample + Render (DX11VideoProcessor.cpp:2143-2200)

    // GetDisplayedImage: a LocalAlloc block = BITMAPINFOHEADER + the back buffer's pixels (:3610-3683); the caller frees it (:3659)
    BYTE *dib = nullptr; unsigned dib_size = 0;
    CHECK(vp->GetDisplayedImage(&dib, &dib_size));
    const BITMAPINFOHEADER *bih = (const BITMAPINFOHEADER *)dib;
    if (!dib || bih->biSize != sizeof(BITMAPINFOHEADER) || bih->biWidth != win_w || bih->biHeight != -win_h || bih->biPlanes != 1 ||
        dib_size != sizeof(BITMAPINFOHEADER) + bih->biSizeImage) { std::fprintf(stderr, "GetDisplayedImage: bad header\n"); return 5; }
    std::printf("DISPLAYED %d %d %d %u\n", (int)bih->biWidth, (int)-bih->biHeight, (int)bih->biBitCount, (unsigned)bih->biSizeImage);
    if (!write_file(argv[14], dib + sizeof(BITMAPINFOHEADER), bih->biSizeImage)) return 73;
    LocalFree(dib);

    // GetCurentImage: the caller sizes the DIB first (VideoRenderer.cpp:979-988: header + source-rect-sized BGRX)
    std::vector<long> snap((sizeof(BITMAPINFOHEADER) + (size_t)w * h * 4 + sizeof(long) - 1) / sizeof(long));
    CHECK(vp->GetCurentImage(snap.data()));
    const BITMAPINFOHEADER *sb = (const BITMAPINFOHEADER *)snap.data();
    std::printf("SNAPSHOT %d %d %d %u\n", (int)sb->biWidth, (int)-sb->biHeight, (int)sb->biBitCount, (unsigned)sb->biSizeImage);
    if (!write_file(argv[15], (const BYTE *)(sb + 1), sb->biSizeImage)) return 73;

    std::wstring info;
    CHECK(vp->GetVPInfo(info));
    std::string narrow(info.begin(), info.end());
    std::printf("VPINFO %s\n", narrow.c_str());
    std::printf("TYPE %d ROTATION %d\n", vp->Type(), vp->GetRotation());
    vp->Flush();
    delete vp;
    return 0;
}
