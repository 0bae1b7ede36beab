// Package gen: the pieces of cmd/parquetgen/gen that change when the generated reader decodes on the GPU.
// Everything else of the generated file (writer, stats, Scan, Next, Levels, the Field constructors) stays as
// cmd/parquetgen/gen/template*.go emit it today, so NewParquetReader / Next / Scan keep their signatures.
//
// To apply: add `"github.com/parsyl/parquet/gpu"` and `sch "github.com/parsyl/parquet/schema"` to the generated imports, replace the `readRowGroup`
// definition of template.go:320-351 with gpuReadRowGroupTpl, append gpuReaderSupportTpl to `tpl`, and append
// the matching Adopt template to each of the six field templates next to the Read method it stands in for
// (template_required.go:25-35, template_optional.go:42-52, template_string.go:41-60,
// template_string_optional.go:62-81, template_bool.go:38-46, template_bool_optional.go:24-33).
// Written without a Go toolchain at hand; not compiled.
package gen

// readRowGroup: same bookkeeping as template.go:320-351, but the loop over the columns only walks the page
// headers; one gpu.Decode per row group does what the Read methods did.
var gpuReadRowGroupTpl = `
func (p *ParquetReader) readRowGroup() error {
	p.rowGroupCursor = 0

	if len(p.rowGroups) == 0 {
		p.rowGroupCount = 0
		return nil
	}

	if err := p.gpuInit(); err != nil {
		return err
	}

	rg := p.rowGroups[0]
	p.fields = getFields(Fields(compressionUnknown))
	p.rowGroupCount = rg.Rows
	p.rowGroupCursor = 0

	var chunks []gpu.Chunk
	var pages []gpu.Page
	var order []gpuField
	for _, col := range rg.Columns() {
		name := strings.Join(col.MetaData.PathInSchema, ".")
		f, ok := p.fields[name]
		if !ok {
			return fmt.Errorf("unknown field: %s", name)
		}
		pgs := p.pages[name]
		if len(pgs) <= p.index {
			break
		}

		pg := pgs[0]
		gf, ok := f.(gpuField)
		if !ok {
			return fmt.Errorf("field %s has no GPU read path", name)
		}
		d, r := gf.MaxLevels()
		first := len(pages)
		var err error
		pages, err = walkPageHeaders(p.r, pg, d > 0, pages)
		if err != nil {
			return fmt.Errorf("unable to read field %s, err: %s", f.Name(), err)
		}
		codec, err := gpuCodec(pg.Codec)
		if err != nil {
			return fmt.Errorf("unable to read field %s, err: %s", f.Name(), err)
		}
		chunks = append(chunks, gpu.Chunk{PhysType: gf.PhysicalType(), MaxDef: d, MaxRep: r,
			FirstPage: int32(first), NPages: int32(len(pages) - first), NumValues: int64(pg.N), Codec: codec})
		order = append(order, gf)
		p.pages[name] = p.pages[name][1:]
	}

	res, err := p.gpu.Decode(p.file, chunks, pages)
	if err != nil {
		return fmt.Errorf("unable to read row group, err: %s", err)
	}
	defer res.Close()
	for i, gf := range order {
		if err := gf.Adopt(res, i); err != nil {
			return fmt.Errorf("unable to read field %s, err: %s", gf.Name(), err)
		}
	}
	p.rowGroups = p.rowGroups[1:]
	return nil
}
`

// Support code appended to the generated file: the reader's GPU state, the WithDevice option (passed through
// the variadic opts NewParquetReader already has, template.go:235,272-276), and the host-side page-header
// walk with the two termination rules of the reference.
var gpuReaderSupportTpl = `
// gpuField is what readRowGroup needs from a generated Field besides the Field interface.
type gpuField interface {
	Field
	MaxLevels() (def, rep int32)         // 0, 0 for the required field categories
	PhysicalType() int32                 // gpu.Int32, gpu.ByteArray, ...
	Adopt(res *gpu.Result, chunk int) error // takes vals / Defs / Reps of one decoded chunk
}

// WithDevice selects the GPU the reader decodes on (default 0).
func WithDevice(id int) func(*ParquetReader) {
	return func(p *ParquetReader) { p.device = id }
}

// added to ParquetReader: device int; gpu *gpu.Ctx; file *gpu.HostBuf

// gpuInit opens the device context and stages the whole file in pinned memory owned by the library, once.
// The reader consumes the file strictly sequentially from byte 4 (template.go:263), so page offsets are
// positions of p.r, which stay valid as offsets into the staged copy.
func (p *ParquetReader) gpuInit() error {
	if p.gpu != nil {
		return nil
	}
	pos, err := p.r.Seek(0, io.SeekCurrent)
	if err != nil {
		return err
	}
	size, err := p.r.Seek(0, io.SeekEnd)
	if err != nil {
		return err
	}
	ctx, err := gpu.New(p.device)
	if err != nil {
		return err
	}
	buf, err := ctx.HostAlloc(int(size))
	if err != nil {
		ctx.Close()
		return err
	}
	if _, err := p.r.Seek(0, io.SeekStart); err != nil {
		return err
	}
	if _, err := io.ReadFull(p.r, buf.Bytes()); err != nil {
		return err
	}
	if _, err := p.r.Seek(pos, io.SeekStart); err != nil {
		return err
	}
	p.gpu, p.file = ctx, buf
	return nil
}

// Close releases the device context (new; the reference reader has nothing to release).
func (p *ParquetReader) Close() {
	if p.file != nil {
		p.file.Free()
	}
	if p.gpu != nil {
		p.gpu.Close()
	}
}

// gpuCodec maps parquet.Page.Codec (sch.CompressionCodec, an int64 enum: parquet.go:33, schema/parquet.go:356-360)
// to the library's codec number.  What the device does not decode is an error, in the words of pageData's default
// arm (fields.go:397-398): GZIP pages are not silently read as uncompressed bytes.
func gpuCodec(c sch.CompressionCodec) (int32, error) {
	switch c {
	case sch.CompressionCodec_UNCOMPRESSED:
		return 0, nil
	case sch.CompressionCodec_SNAPPY:
		return 1, nil
	default:
		return 0, fmt.Errorf("unsupported column chunk codec: %s", c)
	}
}

// walkPageHeaders reads the thrift page headers of one column chunk and skips the payloads, appending one
// gpu.Page per data page.  It ends where the reference's page loops end:
//   required fields: when the NumValues of the pages read reach pg.N            (fields.go:118)
//   optional fields: when header bytes + payload bytes read reach pg.Size       (fields.go:281,314)
func walkPageHeaders(r io.ReadSeeker, pg parquet.Page, optional bool, pages []gpu.Page) ([]gpu.Page, error) {
	var nRead int
	for (optional && nRead < pg.Size) || (!optional && nRead < pg.N) {
		start, err := r.Seek(0, io.SeekCurrent)
		if err != nil {
			return nil, err
		}
		ph, err := parquet.PageHeader(r)
		if err != nil {
			return nil, err
		}
		payload, err := r.Seek(0, io.SeekCurrent)
		if err != nil {
			return nil, err
		}
		n := int64(ph.CompressedPageSize) // what pageData reads for every codec (fields.go:361-396)
		if _, err := r.Seek(n, io.SeekCurrent); err != nil {
			return nil, err
		}
		pages = append(pages, gpu.Page{SrcOff: payload, PayloadLen: int32(n), NumValues: ph.DataPageHeader.NumValues})
		if optional {
			nRead += int(payload-start) + int(n)
		} else {
			nRead += int(ph.DataPageHeader.NumValues)
		}
	}
	return pages, nil
}
`

// Adopt for the required numeric fields (replaces template_required.go:25-35): PLAIN values are raw little
// endian, so the fetched bytes ARE the []T.
var gpuAdoptRequiredNumericTpl = `
func (f *{{.FieldType}}) MaxLevels() (int32, int32) { return 0, 0 }
func (f *{{.FieldType}}) PhysicalType() int32      { return {{gpuPhysType .}} }
func (f *{{.FieldType}}) Adopt(res *gpu.Result, chunk int) error {
	ci, err := res.Info(chunk)
	if err != nil {
		return err
	}
	v := make([]{{.TypeName}}, ci.NValues)
	if len(v) > 0 {
		b := unsafe.Slice((*byte)(unsafe.Pointer(&v[0])), ci.ValueBytes)
		if _, err := res.Fetch(chunk, gpu.OutValues, b); err != nil {
			return err
		}
	}
	f.vals = append(f.vals, v...)
	return nil
}
`

// Adopt for the optional numeric fields (replaces template_optional.go:42-52 and OptionalField.DoRead's level
// handling, fields.go:295-309).
var gpuAdoptOptionalNumericTpl = `
func (f *{{.FieldType}}) MaxLevels() (int32, int32) { return int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep) }
func (f *{{.FieldType}}) PhysicalType() int32      { return {{gpuPhysType .}} }
func (f *{{.FieldType}}) Adopt(res *gpu.Result, chunk int) error {
	defs, reps, err := res.Levels(chunk, int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep))
	if err != nil {
		return err
	}
	f.Defs = append(f.Defs, defs...)
	f.Reps = append(f.Reps, reps...)
	ci, err := res.Info(chunk)
	if err != nil {
		return err
	}
	v := make([]{{removeStar .TypeName}}, ci.NValues)
	if len(v) > 0 {
		b := unsafe.Slice((*byte)(unsafe.Pointer(&v[0])), ci.ValueBytes)
		if _, err := res.Fetch(chunk, gpu.OutValues, b); err != nil {
			return err
		}
	}
	f.vals = append(f.vals, v...)
	return nil
}
`

// Adopt for StringField (replaces template_string.go:41-60).
var gpuAdoptStringTpl = `
func (f *StringField) MaxLevels() (int32, int32) { return 0, 0 }
func (f *StringField) PhysicalType() int32      { return gpu.ByteArray }
func (f *StringField) Adopt(res *gpu.Result, chunk int) error {
	v, err := res.Strings(chunk)
	if err != nil {
		return err
	}
	f.vals = append(f.vals, v...)
	return nil
}
`

// Adopt for StringOptionalField (replaces template_string_optional.go:62-81, whose loop re-counts every level
// per value).
var gpuAdoptStringOptionalTpl = `
func (f *StringOptionalField) MaxLevels() (int32, int32) { return int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep) }
func (f *StringOptionalField) PhysicalType() int32      { return gpu.ByteArray }
func (f *StringOptionalField) Adopt(res *gpu.Result, chunk int) error {
	defs, reps, err := res.Levels(chunk, int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep))
	if err != nil {
		return err
	}
	f.Defs = append(f.Defs, defs...)
	f.Reps = append(f.Reps, reps...)
	v, err := res.Strings(chunk)
	if err != nil {
		return err
	}
	f.vals = append(f.vals, v...)
	return nil
}
`

// Adopt for BoolField (replaces template_bool.go:38-46, which ASSIGNS f.vals) and BoolOptionalField
// (template_bool_optional.go:24-33, which appends): one byte per value, 0 or 1 (parquet.go:455-505).
var gpuAdoptBoolTpl = `
func gpuBools(res *gpu.Result, chunk int) ([]bool, error) {
	ci, err := res.Info(chunk)
	if err != nil {
		return nil, err
	}
	b, err := res.Bytes(chunk, gpu.OutValues, ci.ValueBytes)
	if err != nil {
		return nil, err
	}
	out := make([]bool, len(b))
	for i, x := range b {
		out[i] = x != 0
	}
	return out, nil
}

func (f *BoolField) MaxLevels() (int32, int32) { return 0, 0 }
func (f *BoolField) PhysicalType() int32      { return gpu.Boolean }
func (f *BoolField) Adopt(res *gpu.Result, chunk int) error {
	v, err := gpuBools(res, chunk)
	f.vals = v
	return err
}

func (f *BoolOptionalField) MaxLevels() (int32, int32) { return int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep) }
func (f *BoolOptionalField) PhysicalType() int32      { return gpu.Boolean }
func (f *BoolOptionalField) Adopt(res *gpu.Result, chunk int) error {
	defs, reps, err := res.Levels(chunk, int32(f.MaxLevels.Def), int32(f.MaxLevels.Rep))
	if err != nil {
		return err
	}
	f.Defs = append(f.Defs, defs...)
	f.Reps = append(f.Reps, reps...)
	v, err := gpuBools(res, chunk)
	f.vals = append(f.vals, v...)
	return err
}
`

// gpuPhysType is the template function the numeric Adopt templates call: the parquet physical type of the
// field's Go type (cmd/parquetgen/gen/funcs.go maps the same names for the schema).
func gpuPhysType(typeName string) string {
	switch typeName {
	case "int32", "*int32", "uint32", "*uint32":
		return "gpu.Int32"
	case "int64", "*int64", "uint64", "*uint64":
		return "gpu.Int64"
	case "float32", "*float32":
		return "gpu.Float"
	case "float64", "*float64":
		return "gpu.Double"
	}
	return "-1"
}
