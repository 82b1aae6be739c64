// D3Dcompiler.h — stand-in (TEST INFRASTRUCTURE ONLY; ours): the shapes CompileShader() (Source/Shaders.cpp:29-63) uses.
#pragma once
struct D3D_SHADER_MACRO { const char* Name; const char* Definition; };
struct ID3DBlob {
    virtual void* GetBufferPointer() = 0;
    virtual size_t GetBufferSize() = 0;
    virtual void Release() = 0;
    virtual ~ID3DBlob() {}
};
typedef HRESULT (*pD3DCompile)(const void* pSrcData, size_t SrcDataSize, const char* pSourceName, const D3D_SHADER_MACRO* pDefines,
                               void* pInclude, const char* pEntrypoint, const char* pTarget, UINT Flags1, UINT Flags2,
                               ID3DBlob** ppCode, ID3DBlob** ppErrorMsgs);
