/*
 * glref — headless OpenGL 4.3 core context on Mesa llvmpipe, for running the
 * reference renderer's GLSL shaders on the CPU as a parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path links or loads this.
 *
 * It drives Mesa's software rasteriser directly through the DRI "swrast"
 * loader interface (no X server, GLX, EGL, GLFW or GLEW):
 *   dlopen(swrast_dri.so) -> __driDriverGetExtensions_swrast -> DRI_Core + DRI_SWRast
 *   GL entry points come from libglapi.so.0:_glapi_get_proc_address.
 *
 * The shader sources are NOT part of this repository: callers pass the paths
 * of the reference's own files (src/shaders/*.glsl) and this library compiles
 * them as they are, optionally after one textual find/replace that is used to
 * make the tracer emit primary-hit ids instead of a colour.
 *
 * The exported functions are a thin, generic GL wrapper; the sequence of GL
 * calls that re-creates the reference's frame loop (src/main.cpp:116-279,
 * src/Scene.cpp:25-32,87-153) lives in oracle/glsl_ref.py.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GL_GLEXT_PROTOTYPES 0
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static char g_err[4096];
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

const char *glref_error(void) { return g_err; }

/* ---- swrast loader stubs (the window-system side, which we do not have) ---- */
static void ld_getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void ld_putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void ld_putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void ld_getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 3 },
    .getDrawableInfo = ld_getDrawableInfo,
    .putImage = ld_putImage,
    .getImage = ld_getImage,
    .putImage2 = ld_putImage2,
    .getImage2 = ld_getImage2,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

/* ---- GL entry points ---- */
#define GLFUNCS(X) \
    X(PFNGLGETSTRINGPROC_, GetString) \
    X(PFNGLGETERRORPROC_, GetError) \
    X(PFNGLFINISHPROC_, Finish) \
    X(PFNGLVIEWPORTPROC_, Viewport) \
    X(PFNGLDRAWARRAYSPROC_, DrawArrays) \
    X(PFNGLGENTEXTURESPROC_, GenTextures) \
    X(PFNGLBINDTEXTUREPROC_, BindTexture) \
    X(PFNGLTEXIMAGE2DPROC_, TexImage2D) \
    X(PFNGLTEXPARAMETERIPROC_, TexParameteri) \
    X(PFNGLGETTEXIMAGEPROC_, GetTexImage) \
    X(PFNGLPIXELSTOREIPROC_, PixelStorei) \
    X(PFNGLCLEARPROC_, Clear) \
    X(PFNGLCLEARCOLORPROC_, ClearColor) \
    X(PFNGLACTIVETEXTUREPROC, ActiveTexture) \
    X(PFNGLCREATESHADERPROC, CreateShader) \
    X(PFNGLSHADERSOURCEPROC, ShaderSource) \
    X(PFNGLCOMPILESHADERPROC, CompileShader) \
    X(PFNGLGETSHADERIVPROC, GetShaderiv) \
    X(PFNGLGETSHADERINFOLOGPROC, GetShaderInfoLog) \
    X(PFNGLCREATEPROGRAMPROC, CreateProgram) \
    X(PFNGLATTACHSHADERPROC, AttachShader) \
    X(PFNGLBINDATTRIBLOCATIONPROC, BindAttribLocation) \
    X(PFNGLLINKPROGRAMPROC, LinkProgram) \
    X(PFNGLGETPROGRAMIVPROC, GetProgramiv) \
    X(PFNGLGETPROGRAMINFOLOGPROC, GetProgramInfoLog) \
    X(PFNGLUSEPROGRAMPROC, UseProgram) \
    X(PFNGLGETUNIFORMLOCATIONPROC, GetUniformLocation) \
    X(PFNGLUNIFORM1FPROC, Uniform1f) \
    X(PFNGLUNIFORM2FPROC, Uniform2f) \
    X(PFNGLUNIFORM3FPROC, Uniform3f) \
    X(PFNGLUNIFORM1IPROC, Uniform1i) \
    X(PFNGLUNIFORM1UIPROC, Uniform1ui) \
    X(PFNGLGENVERTEXARRAYSPROC, GenVertexArrays) \
    X(PFNGLBINDVERTEXARRAYPROC, BindVertexArray) \
    X(PFNGLGENBUFFERSPROC, GenBuffers) \
    X(PFNGLBINDBUFFERPROC, BindBuffer) \
    X(PFNGLBUFFERDATAPROC, BufferData) \
    X(PFNGLBINDBUFFERBASEPROC, BindBufferBase) \
    X(PFNGLENABLEVERTEXATTRIBARRAYPROC, EnableVertexAttribArray) \
    X(PFNGLVERTEXATTRIBPOINTERPROC, VertexAttribPointer) \
    X(PFNGLGENFRAMEBUFFERSPROC, GenFramebuffers) \
    X(PFNGLBINDFRAMEBUFFERPROC, BindFramebuffer) \
    X(PFNGLFRAMEBUFFERTEXTURE2DPROC, FramebufferTexture2D) \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, CheckFramebufferStatus) \
    X(PFNGLBLITFRAMEBUFFERPROC, BlitFramebuffer) \
    X(PFNGLTEXBUFFERPROC, TexBuffer) \
    X(PFNGLDELETETEXTURESPROC_, DeleteTextures) \
    X(PFNGLDELETEBUFFERSPROC, DeleteBuffers)

/* GL 1.x prototypes do not have PFN typedefs in every gl.h; define our own. */
typedef const GLubyte *(*PFNGLGETSTRINGPROC_)(GLenum);
typedef GLenum (*PFNGLGETERRORPROC_)(void);
typedef void (*PFNGLFINISHPROC_)(void);
typedef void (*PFNGLVIEWPORTPROC_)(GLint, GLint, GLsizei, GLsizei);
typedef void (*PFNGLDRAWARRAYSPROC_)(GLenum, GLint, GLsizei);
typedef void (*PFNGLGENTEXTURESPROC_)(GLsizei, GLuint *);
typedef void (*PFNGLBINDTEXTUREPROC_)(GLenum, GLuint);
typedef void (*PFNGLTEXIMAGE2DPROC_)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
typedef void (*PFNGLTEXPARAMETERIPROC_)(GLenum, GLenum, GLint);
typedef void (*PFNGLGETTEXIMAGEPROC_)(GLenum, GLint, GLenum, GLenum, void *);
typedef void (*PFNGLPIXELSTOREIPROC_)(GLenum, GLint);
typedef void (*PFNGLCLEARPROC_)(GLbitfield);
typedef void (*PFNGLCLEARCOLORPROC_)(GLfloat, GLfloat, GLfloat, GLfloat);
typedef void (*PFNGLDELETETEXTURESPROC_)(GLsizei, const GLuint *);

#define X(T, N) static T p##N;
GLFUNCS(X)
#undef X

static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_sw;
static __DRIscreen *g_scr;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static GLuint g_vao, g_vbo;
static int g_ready;

static int gl_check(const char *what)
{
    GLenum e = pGetError();
    if (e != GL_NO_ERROR) FAIL("GL error 0x%x after %s", e, what);
    return 0;
}

int glref_init(void)
{
    if (g_ready) return 0;
    void *drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) drv = dlopen("swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) FAIL("dlopen swrast_dri.so: %s", dlerror());
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) FAIL("no __driDriverGetExtensions_swrast");
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_sw = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_sw) FAIL("DRI_Core / DRI_SWRast extension missing");
    if (g_sw->base.version < 4) FAIL("DRI_SWRast version %d < 4", g_sw->base.version);

    const __DRIconfig **configs = NULL;
    g_scr = g_sw->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
    if (!g_scr || !configs || !configs[0]) FAIL("createNewScreen2 failed");
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3 };
    unsigned err = 0;
    g_ctx = g_sw->createContextAttribs(g_scr, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) FAIL("createContextAttribs failed (dri error %u)", err);
    g_draw = g_sw->createNewDrawable(g_scr, configs[0], NULL);
    if (!g_draw) FAIL("createNewDrawable failed");
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) FAIL("bindContext failed");

    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) FAIL("dlopen libglapi.so.0: %s", dlerror());
    void *(*gpa)(const char *) = (void *(*)(const char *))dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) FAIL("no _glapi_get_proc_address");
#define X(T, N) p##N = (T)gpa("gl" #N); if (!p##N) FAIL("missing gl" #N);
    GLFUNCS(X)
#undef X

    /* full-screen quad: the six vertices of src/main.cpp:128-135 */
    static const float quad[12] = { 1, 1, -1, -1, 1, -1, 1, 1, -1, 1, -1, -1 };
    pGenVertexArrays(1, &g_vao);
    pBindVertexArray(g_vao);
    pGenBuffers(1, &g_vbo);
    pBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    pBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    pEnableVertexAttribArray(0);
    pVertexAttribPointer(0, 2, GL_FLOAT, GL_FALSE, 8, (void *)0);
    pPixelStorei(GL_PACK_ALIGNMENT, 1);
    pPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    if (gl_check("init")) return -1;
    g_ready = 1;
    return 0;
}

const char *glref_renderer(void) { return (const char *)pGetString(GL_RENDERER); }
const char *glref_version(void) { return (const char *)pGetString(GL_VERSION); }

static char *read_file(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *s = malloc(n + 1);
    if (fread(s, 1, n, f) != (size_t)n) { fclose(f); free(s); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

static GLuint compile(GLenum type, const char *src)
{
    GLuint sh = pCreateShader(type);
    pShaderSource(sh, 1, &src, NULL);
    pCompileShader(sh);
    GLint ok = 0;
    pGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[3000];
        pGetShaderInfoLog(sh, sizeof log, NULL, log);
        snprintf(g_err, sizeof g_err, "shader compile failed: %s", log);
        return 0;
    }
    return sh;
}

/* Compile + link a program from two files. If `find` is non-NULL the first
 * occurrence of that exact text in the fragment source is replaced by `repl`
 * (it must occur, otherwise the call fails). Returns the program id, 0 on error. */
unsigned glref_program(const char *vs_path, const char *fs_path, const char *find, const char *repl)
{
    char *vs = read_file(vs_path), *fs = read_file(fs_path);
    if (!vs || !fs) { snprintf(g_err, sizeof g_err, "cannot read %s / %s", vs_path, fs_path); return 0; }
    if (find && *find) {
        char *at = strstr(fs, find);
        if (!at) { snprintf(g_err, sizeof g_err, "patch anchor not found in %s", fs_path); return 0; }
        size_t n = strlen(fs) + strlen(repl) + 1;
        char *out = malloc(n);
        size_t head = (size_t)(at - fs);
        memcpy(out, fs, head);
        strcpy(out + head, repl);
        strcat(out, at + strlen(find));
        free(fs);
        fs = out;
    }
    GLuint v = compile(GL_VERTEX_SHADER, vs), f = compile(GL_FRAGMENT_SHADER, fs);
    free(vs); free(fs);
    if (!v || !f) return 0;
    GLuint prog = pCreateProgram();
    pAttachShader(prog, v);
    pAttachShader(prog, f);
    pBindAttribLocation(prog, 0, "position");
    pLinkProgram(prog);
    GLint ok = 0;
    pGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3000];
        pGetProgramInfoLog(prog, sizeof log, NULL, log);
        snprintf(g_err, sizeof g_err, "link failed: %s", log);
        return 0;
    }
    return prog;
}

void glref_use(unsigned prog) { pUseProgram(prog); }

/* Uniform setters act on the program given (it becomes current). Unknown names are ignored,
 * like glUniform* with location -1. */
void glref_uniform1f(unsigned prog, const char *name, float a)
{ pUseProgram(prog); pUniform1f(pGetUniformLocation(prog, name), a); }
void glref_uniform2f(unsigned prog, const char *name, float a, float b)
{ pUseProgram(prog); pUniform2f(pGetUniformLocation(prog, name), a, b); }
void glref_uniform3f(unsigned prog, const char *name, float a, float b, float c)
{ pUseProgram(prog); pUniform3f(pGetUniformLocation(prog, name), a, b, c); }
void glref_uniform1i(unsigned prog, const char *name, int a)
{ pUseProgram(prog); pUniform1i(pGetUniformLocation(prog, name), a); }
void glref_uniform1ui(unsigned prog, const char *name, unsigned a)
{ pUseProgram(prog); pUniform1ui(pGetUniformLocation(prog, name), a); }
int glref_uniform_location(unsigned prog, const char *name) { return pGetUniformLocation(prog, name); }

/* RGBA32F 2-D texture; data may be NULL (then contents are zero-filled here so that runs are
 * deterministic; the reference leaves them undefined). filter: 0 NEAREST, 1 LINEAR. */
unsigned glref_tex2d_rgba32f(int unit, int w, int h, const float *rgba, int linear)
{
    GLuint t;
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_2D, t);
    float *zero = NULL;
    if (!rgba) { zero = calloc((size_t)w * h * 4, sizeof(float)); rgba = zero; }
    pTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, rgba);
    free(zero);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    return t;
}

void glref_tex2d_upload_rgba32f(int unit, unsigned tex, int w, int h, const float *rgba)
{
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_2D, tex);
    pTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, rgba);
}

/* RGB8 noise texture, LINEAR + REPEAT (src/main.cpp:89-114). */
unsigned glref_tex2d_rgb8(int unit, int w, int h, const unsigned char *rgb)
{
    GLuint t;
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_2D, t);
    pTexImage2D(GL_TEXTURE_2D, 0, GL_RGB, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE, rgb);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
    return t;
}

/* RGBA8 cube map, LINEAR + CLAMP_TO_EDGE, faces in GL order +X,-X,+Y,-Y,+Z,-Z
 * (src/main.cpp:45-87). `faces` is 6*size*size*4 bytes. */
unsigned glref_cubemap_rgba8(int unit, int size, const unsigned char *faces)
{
    GLuint t;
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_CUBE_MAP, t);
    for (int i = 0; i < 6; ++i)
        pTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + i, 0, GL_RGBA, size, size, 0, GL_RGBA, GL_UNSIGNED_BYTE,
                    faces + (size_t)i * size * size * 4);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_R, GL_CLAMP_TO_EDGE);
    return t;
}

/* Texture buffer (src/Scene.h:84-105). kind: 0 = R32I, 1 = RGBA32F. */
unsigned glref_texbuffer(int unit, const void *data, long nbytes, int kind)
{
    GLuint b, t;
    pGenBuffers(1, &b);
    pBindBuffer(GL_TEXTURE_BUFFER, b);
    pBufferData(GL_TEXTURE_BUFFER, nbytes, data, GL_STATIC_DRAW);
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_BUFFER, t);
    pTexBuffer(GL_TEXTURE_BUFFER, kind == 0 ? GL_R32I : GL_RGBA32F, b);
    pBindBuffer(GL_TEXTURE_BUFFER, 0);
    return t;
}

/* Shader storage buffer bound to `binding` (src/Scene.h:69-82). */
unsigned glref_ssbo(int binding, const void *data, long nbytes)
{
    GLuint b;
    pGenBuffers(1, &b);
    pBindBuffer(GL_SHADER_STORAGE_BUFFER, b);
    pBufferData(GL_SHADER_STORAGE_BUFFER, nbytes, data, GL_STATIC_DRAW);
    pBindBufferBase(GL_SHADER_STORAGE_BUFFER, binding, b);
    pBindBuffer(GL_SHADER_STORAGE_BUFFER, 0);
    return b;
}

unsigned glref_fbo(unsigned tex)
{
    GLuint f;
    pGenFramebuffers(1, &f);
    pBindFramebuffer(GL_FRAMEBUFFER, f);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    GLenum st = pCheckFramebufferStatus(GL_FRAMEBUFFER);
    pBindFramebuffer(GL_FRAMEBUFFER, 0);
    if (st != GL_FRAMEBUFFER_COMPLETE) { snprintf(g_err, sizeof g_err, "fbo incomplete 0x%x", st); return 0; }
    return f;
}

/* One full-screen-quad draw with `prog` into `fbo` (src/main.cpp:232-233). */
int glref_draw(unsigned prog, unsigned fbo, int w, int h)
{
    pUseProgram(prog);
    pBindFramebuffer(GL_FRAMEBUFFER, fbo);
    pViewport(0, 0, w, h);
    pBindVertexArray(g_vao);
    pDrawArrays(GL_TRIANGLES, 0, 6);
    return gl_check("draw");
}

int glref_blit(unsigned src_fbo, unsigned dst_fbo, int sx0, int sy0, int sx1, int sy1,
               int dx0, int dy0, int dx1, int dy1, int linear)
{
    pBindFramebuffer(GL_READ_FRAMEBUFFER, src_fbo);
    pBindFramebuffer(GL_DRAW_FRAMEBUFFER, dst_fbo);
    pBlitFramebuffer(sx0, sy0, sx1, sy1, dx0, dy0, dx1, dy1, GL_COLOR_BUFFER_BIT, linear ? GL_LINEAR : GL_NEAREST);
    return gl_check("blit");
}

void glref_finish(void) { pFinish(); }

int glref_read_rgba32f(int unit, unsigned tex, float *out)
{
    pFinish();
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_2D, tex);
    pGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, GL_FLOAT, out);
    return gl_check("read rgba32f");
}

/* The Recorder's read-back: GL_RGB / GL_UNSIGNED_BYTE (Include/Recorder.h:93-100). */
int glref_read_rgb8(int unit, unsigned tex, unsigned char *out)
{
    pFinish();
    pActiveTexture(GL_TEXTURE0 + unit);
    pBindTexture(GL_TEXTURE_2D, tex);
    pGetTexImage(GL_TEXTURE_2D, 0, GL_RGB, GL_UNSIGNED_BYTE, out);
    return gl_check("read rgb8");
}
