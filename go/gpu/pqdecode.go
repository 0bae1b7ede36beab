// Package gpu binds libpqdecode.so (include/pqdecode.h): the MI355X decoder for the read path of
// github.com/parsyl/parquet.  One call to Ctx.Decode replaces one readRowGroup of the generated reader
// (cmd/parquetgen/gen/template.go:320-351): every column chunk of the row group goes to the device as one
// batch and comes back as the slices the generated Fields keep (vals, Defs, Reps).
//
// Written without a Go toolchain at hand (the build container has none): it is checked against the header by
// tests/test_go_shim_cpu.py (every C symbol used here is declared there) but has not been compiled.
package gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../parquet_amd/lib -lpqdecode -Wl,-rpath,${SRCDIR}/../../parquet_amd/lib
#include <stdlib.h>
#include "pqdecode.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Physical types (schema/parquet.thrift Type), as the descriptors carry them.
const (
	Boolean   = int32(C.PQD_BOOLEAN)
	Int32     = int32(C.PQD_INT32)
	Int64     = int32(C.PQD_INT64)
	Float     = int32(C.PQD_FLOAT)
	Double    = int32(C.PQD_DOUBLE)
	ByteArray = int32(C.PQD_BYTE_ARRAY)
)

// Output selects one decoded array of a chunk.
type Output C.pqd_output

const (
	OutDefs        = Output(C.PQD_OUT_DEFS)
	OutReps        = Output(C.PQD_OUT_REPS)
	OutValues      = Output(C.PQD_OUT_VALUES)
	OutOffsets     = Output(C.PQD_OUT_OFFSETS)
	OutChars       = Output(C.PQD_OUT_CHARS)
	OutValidity    = Output(C.PQD_OUT_VALIDITY)
	OutPageNonNull = Output(C.PQD_OUT_PAGE_NONNULL)
	OutRowLevelOff = Output(C.PQD_OUT_ROW_LEVEL_OFF)
	OutRowValueOff = Output(C.PQD_OUT_ROW_VALUE_OFF)
	OutValueIndex  = Output(C.PQD_OUT_VALUE_INDEX)
)

// Ctx is one decoder context on one device: one per ParquetReader, not shared between goroutines at the
// same time (the reference reader is single goroutine).
type Ctx struct{ h *C.pqd_ctx }

// New opens a context on `device`.  There is no CPU path: without a usable device it fails.
func New(device int) (*Ctx, error) {
	var h *C.pqd_ctx
	if rc := C.pqd_create(C.int(device), &h); rc != 0 {
		return nil, fmt.Errorf("pqd_create: status %d (no usable MI355X device)", int(rc))
	}
	return &Ctx{h}, nil
}

func (c *Ctx) Close() {
	if c.h != nil {
		C.pqd_destroy(c.h)
		c.h = nil
	}
}

func (c *Ctx) lastError() string { return C.GoString(C.pqd_last_error(c.h)) }

// HostBuf is pinned host memory owned by the library: the file bytes are staged there so that C never keeps
// a Go pointer after a call returns (cgo pointer rules).
type HostBuf struct {
	ctx *Ctx
	P   unsafe.Pointer
	N   int
}

func (c *Ctx) HostAlloc(n int) (*HostBuf, error) {
	p := C.pqd_host_alloc(c.h, C.size_t(n))
	if p == nil {
		return nil, fmt.Errorf("pqd_host_alloc(%d) failed", n)
	}
	return &HostBuf{c, p, n}, nil
}

// Bytes is a Go view of the buffer (valid until Free).
func (b *HostBuf) Bytes() []byte { return unsafe.Slice((*byte)(b.P), b.N) }
func (b *HostBuf) Free() {
	if b.P != nil {
		C.pqd_host_free(b.ctx.h, b.P)
		b.P = nil
	}
}

// Chunk mirrors parquet.Page (parquet.go:28-34, one per column chunk despite its name) plus what the
// generated Field knows: physical type and max levels (fields.go:38-57).
type Chunk struct {
	PhysType, MaxDef, MaxRep int32
	FirstPage, NPages        int32
	NumValues                int64 // parquet.Page.N
	Codec                    int32 // parquet.Page.Codec: 0 uncompressed, 1 snappy
}

// Page is one data page as DoRead sees it (fields.go:114-135, 275-317): where its payload starts (first byte
// after the thrift PageHeader), how many bytes pageData reads, and DataPageHeader.NumValues.
type Page struct {
	SrcOff                int64
	PayloadLen, NumValues int32
}

// Result holds the decoded row group on the device until Close.
type Result struct {
	ctx  *Ctx
	plan *C.pqd_plan
}

// Decode is one readRowGroup: all chunks, all pages, one batch.
func (c *Ctx) Decode(file *HostBuf, chunks []Chunk, pages []Page) (*Result, error) {
	if len(chunks) == 0 {
		return &Result{c, nil}, nil
	}
	cc := make([]C.pqd_chunk_desc, len(chunks))
	for i, ch := range chunks {
		cc[i] = C.pqd_chunk_desc{phys_type: C.int32_t(ch.PhysType), max_def: C.int32_t(ch.MaxDef),
			max_rep: C.int32_t(ch.MaxRep), first_page: C.int32_t(ch.FirstPage), n_pages: C.int32_t(ch.NPages),
			column: C.int32_t(i), num_values: C.int64_t(ch.NumValues), codec: C.int32_t(ch.Codec)}
	}
	pp := make([]C.pqd_page_desc, len(pages)+1) // +1: &pp[0] must exist for a row group without pages
	for i, p := range pages {
		// rep_bytes / def_bytes = -1: the library reads the level block lengths from the host copy (rle.go:161-164)
		pp[i] = C.pqd_page_desc{src_off: C.int64_t(p.SrcOff), payload_len: C.int32_t(p.PayloadLen),
			num_values: C.int32_t(p.NumValues), rep_bytes: -1, def_bytes: -1}
	}
	var plan *C.pqd_plan
	rc := C.pqd_decode(c.h, &cc[0], C.size_t(len(cc)), &pp[0], C.size_t(len(pages)), file.P, C.size_t(file.N), &plan)
	if rc != 0 {
		msg := c.lastError()
		if plan != nil {
			C.pqd_plan_destroy(plan)
		}
		return nil, fmt.Errorf("pqd_decode: status %d: %s", int(rc), msg)
	}
	return &Result{c, plan}, nil
}

// Info is pqd_chunk_info of one chunk.
type Info struct{ NLevels, NValues, ValueBytes, CharsBytes, NRows, NPages int64 }

func (r *Result) Info(chunk int) (Info, error) {
	if r.plan == nil {
		return Info{}, fmt.Errorf("no decoded row group (empty batch or closed result)")
	}
	var ci C.pqd_chunk_info
	if rc := C.pqd_plan_chunk_info(r.plan, C.size_t(chunk), &ci); rc != 0 {
		return Info{}, fmt.Errorf("pqd_plan_chunk_info: status %d", int(rc))
	}
	return Info{int64(ci.n_levels), int64(ci.n_values), int64(ci.value_bytes), int64(ci.chars_bytes), int64(ci.n_rows), int64(ci.n_pages)}, nil
}

// Fetch copies one decoded array of one chunk into dst and returns the bytes copied.
func (r *Result) Fetch(chunk int, which Output, dst []byte) (int, error) {
	if r.plan == nil {
		return 0, fmt.Errorf("no decoded row group (empty batch or closed result)")
	}
	if len(dst) == 0 {
		return 0, nil
	}
	n := C.pqd_plan_fetch(r.plan, C.size_t(chunk), C.pqd_output(which), unsafe.Pointer(&dst[0]), C.size_t(len(dst)))
	if n < 0 {
		return 0, fmt.Errorf("pqd_plan_fetch: status %d: %s", int(n), r.ctx.lastError())
	}
	return int(n), nil
}

// Bytes fetches an array of n bytes.
func (r *Result) Bytes(chunk int, which Output, n int64) ([]byte, error) {
	out := make([]byte, n)
	got, err := r.Fetch(chunk, which, out)
	return out[:got], err
}

// Levels returns Defs and Reps of a chunk (nil where the column has none): fields.go:300,308.
func (r *Result) Levels(chunk int, maxDef, maxRep int32) (defs, reps []uint8, err error) {
	ci, err := r.Info(chunk)
	if err != nil {
		return nil, nil, err
	}
	if maxDef > 0 {
		if defs, err = r.Bytes(chunk, OutDefs, ci.NLevels); err != nil {
			return nil, nil, err
		}
	}
	if maxRep > 0 {
		if reps, err = r.Bytes(chunk, OutReps, ci.NLevels); err != nil {
			return nil, nil, err
		}
	}
	return defs, reps, nil
}

// Strings rebuilds the []string of a BYTE_ARRAY chunk from its offsets and chars (template_string.go:47-58).
func (r *Result) Strings(chunk int) ([]string, error) {
	ci, err := r.Info(chunk)
	if err != nil {
		return nil, err
	}
	ob, err := r.Bytes(chunk, OutOffsets, (ci.NValues+1)*4)
	if err != nil {
		return nil, err
	}
	chars, err := r.Bytes(chunk, OutChars, ci.CharsBytes)
	if err != nil {
		return nil, err
	}
	if int64(len(ob)) < 4*(ci.NValues+1) {
		return nil, fmt.Errorf("string chunk %d: %d offset bytes for %d values", chunk, len(ob), ci.NValues)
	}
	all := string(chars) // one copy; the values below share it
	out := make([]string, ci.NValues)
	offs := unsafe.Slice((*int32)(unsafe.Pointer(&ob[0])), ci.NValues+1)
	for i := range out {
		lo, hi := int64(offs[i]), int64(offs[i+1])
		if lo < 0 || hi < lo || hi > int64(len(all)) {
			return nil, fmt.Errorf("string chunk %d: offsets of value %d out of order or outside the %d chars", chunk, i, len(all))
		}
		out[i] = all[lo:hi]
	}
	return out, nil
}

// Scattered returns the values of an optional chunk at their level slots (pqd_plan_scatter): slotBytes = 4 or 8 for
// the fixed-width types, 1 for bools; nulls read as zero.  For a BYTE_ARRAY chunk pass slotBytes = 4 and get the
// NLevels+1 slot offsets into OutChars (a null has length 0).  The dense Values plus Defs stay what Scan consumes
// (performance/parquet.go:104-113); this is for callers that want one entry per row.
func (r *Result) Scattered(chunk int, slotBytes int, strings bool) ([]byte, error) {
	if r.plan == nil {
		return nil, fmt.Errorf("no decoded row group (empty batch or closed result)")
	}
	ci, err := r.Info(chunk)
	if err != nil {
		return nil, err
	}
	n := ci.NLevels
	if strings {
		n++
	}
	size := C.size_t(n * int64(slotBytes))
	if size == 0 {
		return nil, nil
	}
	dev := C.pqd_device_alloc(r.ctx.h, size)
	if dev == nil {
		return nil, fmt.Errorf("pqd_device_alloc of %d bytes failed", int64(size))
	}
	defer C.pqd_device_free(r.ctx.h, dev)
	if rc := C.pqd_plan_scatter(r.plan, C.size_t(chunk), dev, size); rc != 0 {
		return nil, fmt.Errorf("pqd_plan_scatter: status %d: %s", int(rc), r.ctx.lastError())
	}
	if rc := C.pqd_plan_sync(r.plan); rc != 0 {
		return nil, fmt.Errorf("pqd_plan_sync: status %d: %s", int(rc), r.ctx.lastError())
	}
	out := make([]byte, int(size))
	if rc := C.pqd_copy_to_host(r.ctx.h, unsafe.Pointer(&out[0]), dev, size); rc != 0 {
		return nil, fmt.Errorf("pqd_copy_to_host: status %d: %s", int(rc), r.ctx.lastError())
	}
	return out, nil
}

func (r *Result) Close() {
	if r.plan != nil {
		C.pqd_plan_destroy(r.plan)
		r.plan = nil
	}
}
