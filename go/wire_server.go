// wire_server.go — the payload stage behind the gRPC server: the handler registered in V1_ServiceDesc's place never unmarshals, it hands the
// bytes of the GetRateLimitsReq to guber_wire_pool_get_rate_limits and returns the bytes of the GetRateLimitsResp (include/guber_wire.h,
// INTEGRATION.md section 3g).  What it replaces: _V1_GetRateLimits_Handler (gubernator_grpc.pb.go:111-127) + V1Instance.GetRateLimits
// (gubernator.go:183-306) for requests this instance owns; _PeersV1_GetPeerRateLimits_Handler (peers_grpc.pb.go:109) likewise.
//
// NOT compiled in this repository (no Go toolchain in the build image): the calls below are made, in the same order and with the same
// arguments, by tests/hostsim/abi_c99.c (wire_pool_sequence) — compiled as C99 against the public headers and run on the GPU.
package gubernator

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../gubernator_amd -lguber_hip
#include "guber_gpu.h"
#include "guber_wire.h"
*/
import "C"

import (
	"context"
	"unsafe"

	"google.golang.org/grpc"
	"google.golang.org/grpc/codes"
	"google.golang.org/grpc/status"
)

// rawMessage is a message that IS its bytes; rawCodec leaves it alone (grpc.ForceServerCodec(rawCodec{}) on the servers of daemon.go:119-144).
type rawMessage struct{ b []byte }
type rawCodec struct{}

func (rawCodec) Marshal(v interface{}) ([]byte, error)   { return v.(*rawMessage).b, nil }
func (rawCodec) Unmarshal(d []byte, v interface{}) error { v.(*rawMessage).b = d; return nil }
func (rawCodec) Name() string                             { return "proto" }

// WireServer owns the payload stage of ONE device.
type WireServer struct {
	pool *C.guber_wire_pool_t
}

// NewWireServer: the engines of the device (GPUWorkerPool's shards), the placement's rule (nil with one table), the defaults of
// guber_wire_pool_config_t (twelve stages of 49 152 items, BatchWait 500 us, 1000 requests per RPC).
func NewWireServer(engines []*C.guber_engine_t, rule *C.struct_guber_route_rule) (*WireServer, error) {
	s := &WireServer{}
	if rc := C.guber_wire_pool_create(&engines[0], C.uint32_t(len(engines)), rule, nil, &s.pool); rc != C.GUBER_OK {
		return nil, status.Errorf(codes.Internal, "guber_wire_pool_create: %s", C.GoString(C.guber_last_error()))
	}
	return s, nil
}

// Close: no call may be in flight or arrive any more (the gRPC servers have been stopped: daemon.go Close).
func (s *WireServer) Close() { C.guber_wire_pool_destroy(s.pool) }

// call hands one serialized message over and returns the serialized answer; wrap: gubernator.go:250-255 (client RPC) or bare texts (peer RPC).
func (s *WireServer) call(in []byte, wrap C.int) ([]byte, error) {
	if len(in) == 0 {
		return nil, nil // no requests: an empty GetRateLimitsResp
	}
	p := (*C.uint8_t)(unsafe.Pointer(&in[0]))
	out := make([]byte, int(C.guber_wire_pool_response_bound(p, C.size_t(len(in)))))
	var n C.size_t
	// blocks (a cgo call: the goroutine keeps its thread) until the stage the payload joined has been through the GPU
	rc := C.guber_wire_pool_get_rate_limits(s.pool, p, C.size_t(len(in)), 1, wrap, (*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(len(out)), &n)
	switch rc {
	case C.GUBER_OK:
		return out[:int(n)], nil
	case C.GUBER_E_WIRE_TOO_LARGE: // gubernator.go:189-193
		return nil, status.Errorf(codes.OutOfRange, "Requests.RateLimits list too large; max size is '%d'", maxBatchSize)
	case C.GUBER_E_WIRE_MALFORMED: // what protobuf-go's Unmarshal failure becomes in grpc-go
		return nil, status.Error(codes.Internal, "grpc: error unmarshalling request")
	default:
		return nil, status.Errorf(codes.Internal, "guber_wire_pool_get_rate_limits: %s", C.GoString(C.guber_last_error()))
	}
}

// The two method handlers, with the signature grpc.MethodDesc.Handler wants (gubernator_grpc.pb.go:150-165, peers_grpc.pb.go:148-163).
func (s *WireServer) getRateLimits(_ interface{}, _ context.Context, dec func(interface{}) error, _ grpc.UnaryServerInterceptor) (interface{}, error) {
	in := new(rawMessage)
	if err := dec(in); err != nil {
		return nil, err
	}
	out, err := s.call(in.b, 1)
	return &rawMessage{out}, err
}

func (s *WireServer) getPeerRateLimits(_ interface{}, _ context.Context, dec func(interface{}) error, _ grpc.UnaryServerInterceptor) (interface{}, error) {
	in := new(rawMessage)
	if err := dec(in); err != nil {
		return nil, err
	}
	out, err := s.call(in.b, 0)
	return &rawMessage{out}, err
}

// ServiceDescs: V1_ServiceDesc / PeersV1_ServiceDesc with the two hot methods replaced (HealthCheck, UpdatePeerGlobals keep the generated handlers).
func (s *WireServer) ServiceDescs() (grpc.ServiceDesc, grpc.ServiceDesc) {
	v1, peers := V1_ServiceDesc, PeersV1_ServiceDesc
	v1.Methods = append([]grpc.MethodDesc(nil), v1.Methods...)
	peers.Methods = append([]grpc.MethodDesc(nil), peers.Methods...)
	for i := range v1.Methods {
		if v1.Methods[i].MethodName == "GetRateLimits" {
			v1.Methods[i].Handler = s.getRateLimits
		}
	}
	for i := range peers.Methods {
		if peers.Methods[i].MethodName == "GetPeerRateLimits" {
			peers.Methods[i].Handler = s.getPeerRateLimits
		}
	}
	return v1, peers
}
